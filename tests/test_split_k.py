"""Split-K plans of the grouped GEMM (``npc._split_k``, round 5): the chain of a C block cut into parts that are tasks of their own,
partial blocks summed by ``tpa_lincomb_batch``.  Checked here: the cut links tile every original link exactly once, the result
equals the dense contraction (real / complex, leading and trailing contracted legs), and the replayed Lanczos program with split
plans gives the run of the unsplit program.  The reference has no counterpart (np.tensordot per block pair,
np_conserved.py:3612 ``tensordot`` -> ``_tensordot_worker`` :3535); the oracle is dense numpy as in its own tests."""
import numpy as np
import pytest

from tenpy_amd.algorithms import mps_common
from tenpy_amd.linalg import krylov_based as kb
from tenpy_amd.linalg import np_conserved as npc
from tenpy_amd.linalg.charges import ChargeInfo, LegCharge
from test_heff import _engine
from test_npc_random import _rand


@pytest.fixture
def split_everything(monkeypatch):
    """Every plan built inside the test is split (up to 4 parts, parts of >= 8 contracted indices)."""
    monkeypatch.setattr(npc, 'GEMM_SPLIT_K', (4, 1 << 30, 8, 1 << 30))
    saved = dict(npc._plan_cache)
    npc._plan_cache.clear()
    yield
    npc._plan_cache.clear()
    npc._plan_cache.update(saved)


def _check_tiling(plan):
    """Every (link, k) of the plan appears in exactly one part, with the operand offsets shifted along the contracted index."""
    sk = plan.sk
    tasks, links = plan.tasks_host, plan.links_host
    it = iter(range(len(sk.tasks_host)))
    n_split = 0
    for t, job in zip(range(len(tasks)), sk.jobs_host):
        assert tuple(job[:4]) == tuple(tasks[t, :4])
        lb, lc = tasks[t, 4], tasks[t, 5]
        want = [(int(l[0]) + k * int(l[4]), int(l[1]) + k * int(l[5])) for l in links[lb:lb + lc] for k in range(int(l[2]))]
        got = []
        n_split += job[5] > 1
        for p in range(job[5]):
            nt = sk.tasks_host[next(it)]
            assert (nt[1], nt[2], nt[3]) == (tasks[t, 1], tasks[t, 2], tasks[t, 2]) and nt[5] >= 1
            assert tuple(sk.terms_host[job[4] + p][:2]) == (nt[0], nt[2])
            for l in sk.links_host[nt[4]:nt[4] + nt[5]]:
                assert l[2] >= 1
                orig = links[lb:lb + lc]
                assert any(np.array_equal(l[3:], o[3:]) for o in orig)       # strides and flags are those of the link it was cut from
                got += [(int(l[0]) + k * int(l[4]), int(l[1]) + k * int(l[5])) for k in range(int(l[2]))]
        assert got == want
    return n_split


@pytest.mark.parametrize("cplx", [False, True])
def test_split_plans_equal_dense(backend, split_everything, cplx):
    rng = np.random.RandomState(5 + cplx)
    ch = ChargeInfo([1])
    big = LegCharge.from_qflat(ch, np.sort(rng.randint(-1, 2, size=90)).reshape(-1, 1), 1).bunch()[1]
    mid = LegCharge.from_qflat(ch, np.sort(rng.randint(-1, 2, size=70)).reshape(-1, 1), 1).bunch()[1]
    w = LegCharge.from_qflat(ch, [[0], [0], [1]], 1)
    small = LegCharge.from_qflat(ch, np.sort(rng.randint(-2, 3, size=20)).reshape(-1, 1), -1).bunch()[1]
    a = _rand(rng, [small, w, big], cplx).iset_leg_labels(['a', 'w', 'k'])
    b = _rand(rng, [big.conj(), w.conj(), mid], cplx).iset_leg_labels(['k*', 'w*', 'b'])
    A, B = a.to_ndarray(), b.to_ndarray()
    n_split = 0
    for axes, dense in [((['w', 'k'], ['w*', 'k*']), np.tensordot(A, B, axes=([1, 2], [1, 0]))),      # trailing legs of a, leading of b
                        ((['k'], ['k*']), np.tensordot(A, B, axes=([2], [0]))),
                        (None, None)]:
        if dense is None:       # leading leg of the first operand, trailing of the second
            a2, b2 = a.transpose(['k', 'w', 'a']), b.transpose(['b', 'w*', 'k*'])
            plan, x, y = npc.plan_tensordot(a2, b2, axes=(['k'], ['k*']))
            assert x is a2 and y is b2
            r = plan.apply(x, y)
            dense = np.tensordot(A.transpose(2, 1, 0), B.transpose(2, 1, 0), axes=([0], [2]))
        else:
            plan, x, y = npc.plan_tensordot(a, b, axes=axes)
            r = plan.apply(x, y)
        assert plan.sk is not None
        n_split += _check_tiling(plan)
        r.test_sanity()
        np.testing.assert_allclose(r.to_ndarray(), dense, rtol=0, atol=1e-12)
    assert n_split > 0


def test_knob_limits(backend, monkeypatch):
    rng = np.random.RandomState(9)
    ch = ChargeInfo([1])
    leg = LegCharge.from_qflat(ch, np.sort(rng.randint(-1, 2, size=60)).reshape(-1, 1), 1).bunch()[1]
    a = _rand(rng, [leg, leg.conj()], False)
    npc._plan_cache.clear()
    monkeypatch.setattr(npc, 'GEMM_SPLIT_K', (0,))                # off
    assert npc.plan_tensordot(a, a, axes=1)[0].sk is None
    npc._plan_cache.clear()
    monkeypatch.setattr(npc, 'GEMM_SPLIT_K', (4, 1 << 30, 8, 1))  # launches of more than one tile are left alone
    assert npc.plan_tensordot(a, a, axes=1)[0].n_tiles > 1 and npc.plan_tensordot(a, a, axes=1)[0].sk is None
    npc._plan_cache.clear()
    monkeypatch.setattr(npc, 'GEMM_SPLIT_K', (4, 1 << 30, 8, 1 << 30))
    assert npc.plan_tensordot(a, a, axes=1)[0].sk is not None
    npc._plan_cache.clear()
    monkeypatch.setattr(npc, 'GEMM_SPLIT_K', (4, 1, 8))           # target of one tile: nothing is "too few tiles"
    assert npc.plan_tensordot(a, a, axes=1)[0].sk is None
    npc._plan_cache.clear()
    monkeypatch.setattr(npc, 'GEMM_SPLIT_K', (4, 2048, 1000))     # parts would be shorter than the minimum
    assert npc.plan_tensordot(a, a, axes=1)[0].sk is None
    npc._plan_cache.clear()


@pytest.mark.parametrize("factored", [False, True])
def test_native_lanczos_with_split_plans(backend, monkeypatch, factored):
    eng = _engine('xxz')
    i0 = eng.psi.L // 2 - 1
    tensors = (eng.env.get_LP(i0), eng.env.get_RP(i0 + 1), eng.H.get_W(i0), eng.H.get_W(i0 + 1))
    res = {}
    saved = dict(npc._plan_cache)
    try:
        for knob in ((0,), (3, 1 << 30, 4, 1 << 30)):
            npc._plan_cache.clear()
            monkeypatch.setattr(npc, 'GEMM_SPLIT_K', knob)
            H = mps_common.TwoSiteH(None, i0, tensors=tensors, factored=factored)
            theta = H.combine_theta(eng.psi.get_theta(i0, n=2))
            monkeypatch.setattr(kb, 'NATIVE', True)
            lz = kb.LanczosGroundState(H, theta, {'N_min': 6, 'N_max': 6})
            prog = lz._native_program()
            assert prog is not None
            ops = prog[0]
            res[knob[0]] = (lz.run(), np.asarray(ops), H.matvec(theta))
    finally:
        npc._plan_cache.clear()
        npc._plan_cache.update(saved)
    (E0, v0, N0), ops0, hv0 = res[0]
    (E1, v1, N1), ops1, hv1 = res[3]
    assert len(ops1) > len(ops0) and np.any((ops1[:, 0] == 1) & (ops1[:, 1] == 1)), "no reduction op: the plans were not split"
    assert N0 == N1 == 6
    assert abs(E1 - E0) <= 1e-12 * max(1., abs(E0))
    assert abs(npc.inner(v0, v1, axes='range', do_conj=True) - 1.) < 1e-12
    assert npc.norm(hv0 - hv1) <= 1e-13 * npc.norm(hv0)


def test_default_knob_on_a_chi2048_structure(backend):
    """The tables of the default knob on the block structure of the chi = 2048 Heisenberg matvec (synthetic Sz sectors, no arithmetic): step 1
    (thousands of tiles) is left alone, step 2 (hundreds of tiles, chains over the MPO index and the bond sectors) is cut into <= 4 parts per block,
    every part covers >= 128 contracted indices, cuts inside a link fall on whole k-tiles, and the parts tile the chains exactly."""
    if backend != 'mock':
        pytest.skip("table check only: the emulated device is enough")
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'scripts'))
    from gemm_bench import sectors
    from tenpy_amd.models.spin_chains import xxz_chain_mpo
    assert npc.GEMM_SPLIT_K[0] >= 2, "the default knob is on"
    chi = 2048
    H = xxz_chain_mpo(8, 1., 1., 0.)
    W0, W1 = H.get_W(3), H.get_W(4)
    q, n = sectors(chi)
    bond = LegCharge.from_qind(W0.chinfo, np.concatenate([[0], np.cumsum(n)]), q.reshape(-1, 1), qconj=+1)
    zeros = lambda sh: np.zeros(sh)
    LP = npc.Array.from_func(zeros, [bond, W0.get_leg('wL').conj(), bond.conj()], labels=['vR*', 'wR', 'vR'])
    RP = npc.Array.from_func(zeros, [bond, W1.get_leg('wR').conj(), bond.conj()], labels=['vL', 'wL', 'vL*'])
    p = W0.get_leg('p')
    theta = npc.Array.from_func(zeros, [bond, p, p, bond.conj()], labels=['vL', 'p0', 'p1', 'vR'])
    eff = mps_common.TwoSiteH(None, 3, tensors=(LP, RP, W0, W1))
    assert eff.factored
    prog = eff.matvec_program(theta)
    assert prog is not None
    p1, p2 = prog[2]
    assert p1.n_tiles > 1024 and p1.sk is None
    sk = p2.sk
    assert 64 < p2.n_tiles <= 1024 and sk is not None
    assert sk.parts.max() <= 4 and sk.parts.min() >= 1 and sk.n_tiles > 2 * p2.n_tiles
    tasks, links = p2.tasks_host, p2.links_host
    # coverage: per original link, the pieces cut from it are disjoint, in order, and add up to its k
    key = lambda l: (int(l[3]), int(l[4]), int(l[5]), int(l[6]), int(l[7]))
    it = 0
    for t in range(len(tasks)):
        lb, lc = int(tasks[t, 4]), int(tasks[t, 5])
        orig = links[lb:lb + lc]
        pos = 0                  # walks through the original chain
        done = 0                 # contracted indices of orig[pos] covered so far
        for _ in range(int(sk.parts[t])):
            nt = sk.tasks_host[it]
            it += 1
            part_k = 0
            for l in sk.links_host[int(nt[4]):int(nt[4] + nt[5])]:
                o = orig[pos]
                assert key(l) == key(o)
                assert int(l[0]) == int(o[0]) + done * int(o[4]) and int(l[1]) == int(o[1]) + done * int(o[5])
                if done:
                    assert done % npc.GEMM_K_TILE == 0, "a cut inside a link falls on a whole k-tile"
                done += int(l[2])
                part_k += int(l[2])
                assert done <= int(o[2])
                if done == int(o[2]):
                    pos, done = pos + 1, 0
            assert part_k >= 100 or sk.parts[t] == 1, "no crumbs: parts are cut at ~ K / parts >= 128, snapped by at most half a k-tile"
        assert pos == lc and done == 0
    assert it == len(sk.tasks_host)
    # the launch program carries the reduction as a kind-1 op with cfg = 1 right after the split GEMM
    ops = np.asarray(prog[0])
    k = [i for i in range(len(ops)) if ops[i, 0] == 1 and ops[i, 1] == 1]
    assert len(k) == 1 and ops[k[0] - 1, 0] == 0 and ops[k[0] - 1, 5] == sk.n_tiles and ops[k[0], 8] == -2
