"""The C-ABI library loads on a machine without a GPU and exports every symbol include/tenpy_amd.h declares;
the host-only entry points work; device entry points are never called here."""
import ctypes
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'tenpy_amd.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(tpa_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_exported():
    from tenpy_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "symbol %s declared in include/tenpy_amd.h but not exported" % n
    # and the python binding declares a signature for each of them
    assert set(names) <= set(_lib.exported_symbols()), set(names) - set(_lib.exported_symbols())
    assert lib.tpa_version() >= 100


def test_plan_tensordot_host():
    """planner output = brute force enumeration; result rows lexsorted (last leg most significant)."""
    from tenpy_amd.linalg.np_conserved import _plan_host
    rng = np.random.default_rng(5)
    for (ra, rb, nc) in [(3, 3, 1), (3, 4, 2), (2, 2, 1), (1, 3, 1), (2, 2, 0)]:
        nblk = 4
        def rand_q(rank, n):
            q = np.unique(rng.integers(0, nblk, size=(n, rank)), axis=0)
            return q[rng.permutation(len(q))]
        aq, bq = rand_q(ra, 15), rand_q(rb, 15)
        res_q, gemm = _plan_host(aq, bq, nc, [nblk] * nc)
        ka = ra - nc
        want = {}
        for i, x in enumerate(aq):
            for j, y in enumerate(bq):
                if tuple(x[ka:]) == tuple(y[:nc]):
                    want.setdefault(tuple(x[:ka]) + tuple(y[nc:]), set()).add((i, j))
        rows = np.array(sorted(want.keys(), key=lambda t: t[::-1]), dtype=np.int64).reshape(len(want), ka + rb - nc)
        np.testing.assert_array_equal(res_q, rows)
        got = {}
        for r, i, j in gemm:
            got.setdefault(tuple(res_q[r]), set()).add((int(i), int(j)))
        assert got == want
        assert np.all(np.diff(gemm[:, 0]) >= 0)


def test_tile_shape():
    from tenpy_amd import _lib
    lib = _lib.load()
    bm, bn = ctypes.c_int(), ctypes.c_int()
    for code in (0, 1):
        for cfg in (0, 1):
            assert lib.tpa_gemm_tile_shape(code, cfg, ctypes.byref(bm), ctypes.byref(bn)) == 0
            assert bm.value % 16 == 0 and bn.value % 16 == 0
