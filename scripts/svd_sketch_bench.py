"""Dev harness (round 5): the stale basis as a SKETCH (tenpy_amd/linalg/_svd_warm.py::svd_blocks_sketch) against the cold path
(pivoted QR + Jacobi) on the dumped chi=2048 theta.  The state's drift between two visits of a bond is modelled by small random
rotations on both sides (rank-preserving: A' = (1 + eps K1) A (1 + eps K2), K antisymmetric) plus a few new directions.

    python scripts/svd_sketch_bench.py [theta.npz]     env: EPS (1e-9), NEW (5 new directions), REPS (3), SIDE (R), PROFILE=1 (stage times)
"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tenpy_amd.linalg import _device as dev
from tenpy_amd.linalg import _svd_warm as sw
from tenpy_amd.linalg import np_conserved as npc
from tenpy_amd.linalg.charges import ChargeInfo, LegCharge

path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), 'data', 'theta_chi2048_sat.npz')
d = np.load(path)
blocks = [np.ascontiguousarray(d[k]) for k in d.files]
ch = ChargeInfo([1])
ms, ns = [b.shape[0] for b in blocks], [b.shape[1] for b in blocks]
legL = LegCharge.from_qind(ch, np.concatenate([[0], np.cumsum(ms)]), np.arange(len(ms))[:, None], 1)
legR = LegCharge.from_qind(ch, np.concatenate([[0], np.cumsum(ns)]), np.arange(len(ns))[:, None], -1)
EPS, NEW, REPS, SIDE = float(os.environ.get('EPS', 1e-9)), int(os.environ.get('NEW', 5)), int(os.environ.get('REPS', 3)), os.environ.get('SIDE', 'R')
rng = np.random.RandomState(0)


def array_of(blks):
    a = npc.Array([legL, legR], np.float64)
    a._set_blocks(np.stack([np.arange(len(ms)), np.arange(len(ms))], axis=1), qdata_sorted=True)
    a._arena = dev.to_device(np.concatenate([b.reshape(-1) for b in blks]))
    return a


def drift(blks):
    out = []
    for b in blks:
        m, n = b.shape
        k1, k2 = rng.standard_normal((m, m)) / np.sqrt(m), rng.standard_normal((n, n)) / np.sqrt(n)
        x = rng.standard_normal((m, NEW)) @ rng.standard_normal((NEW, n)) if NEW else 0.
        nb = b + EPS * ((k1 - k1.T) @ b + b @ (k2 - k2.T))
        if NEW:
            nb = nb + EPS * np.linalg.norm(b) / np.linalg.norm(x) * x
        out.append(np.ascontiguousarray(nb))
    return out


def timed_svd(a, hint):
    for k in list(sw.stats):
        sw.stats[k] = 0
    npc.svd_engine_floor = True
    npc.svd_hint = hint
    gpu = torch.cuda.is_available()
    if gpu:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
    t0 = time.time()
    if gpu:
        e0.record()
    U, S, VH = npc.svd(a)
    if gpu:
        e1.record()
        torch.cuda.synchronize()
    return U, S, VH, 1e3 * (time.time() - t0), e0.elapsed_time(e1) if gpu else 0.


def call_log(n=6):
    """The library's own record of its last tpa_svd_batch calls: (largest block min, max, blocks, sweeps, pivoted QR, switches, wall us, rc)."""
    import ctypes
    buf = (ctypes.c_int64 * (8 * 64))()
    got = dev.lib().tpa_svd_call_log(buf, 64, 1)
    return [list(buf[8 * i:8 * i + 8]) for i in range(got)][-n:]


def check(blks, U, S, VH):
    Ud, Vd = U.to_ndarray(), VH.to_ndarray()
    es, off, r0, c0 = 0., 0, 0, 0
    for b in blks:
        m, n = b.shape
        k = min(m, n)
        ref = np.linalg.svd(b, compute_uv=False)
        es = max(es, np.abs(np.sort(S[off:off + k])[::-1] - ref).max() / ref.max())
        off += k
    A = np.zeros((sum(ms), sum(ns)))
    r0 = c0 = 0
    for b in blks:
        A[r0:r0 + b.shape[0], c0:c0 + b.shape[1]] = b
        r0 += b.shape[0]
        c0 += b.shape[1]
    keep = S > 1e-14 * S.max()
    return dict(sv_err=es, recon=float(np.abs((Ud * S) @ Vd - A).max() / np.abs(A).max()),
                iso=float(max(np.abs(Ud[:, keep].T @ Ud[:, keep] - np.eye(keep.sum())).max(), np.abs(Vd[keep] @ Vd[keep].T - np.eye(keep.sum())).max())))


cur = blocks
a = array_of(cur)
_, _, _, w, ev = timed_svd(a, ('bench', SIDE))
print("call 0 (cold, stores the basis): %.2f ms (events %.2f)" % (w, ev), dict((k, v) for k, v in sw.stats.items() if v and not k.startswith('t_')), flush=True)
for rep in range(REPS):
    cur = drift(cur)
    a = array_of(cur)
    sw.PROFILE = False
    U, S, VH, w, ev = timed_svd(a, ('bench', SIDE))
    kind = 'sketch' if sw.stats.get('sketch_calls') else ('warm' if sw.stats['warm_calls'] else 'cold')
    print("drift %d: %s %.2f ms (events %.2f) sweeps %d e_rel(warm) %.1e e_rel(sketch) %.1e" % (
        rep, kind, w, ev, sw.stats.get('sketch_sweeps', 0) + sw.stats['warm_sweeps'] + sw.stats['cold_sweeps'], sw.stats.get('e_rel_last', 0.),
        sw.stats.get('sk_e_rel_last', 0.)), check(cur, U, S, VH) if os.environ.get('CHECK', '1') != '0' else '', flush=True)
    print("         tpa_svd_batch calls of that npc.svd:", call_log(), "rows cut", sw.stats.get('sk_rows_cut'), flush=True)
    _, _, _, w2, ev2 = timed_svd(a, None)
    print("         the same matrix cold (no hint): %.2f ms (events %.2f) sweeps %d" % (w2, ev2, sw.stats['cold_sweeps']), call_log(2), flush=True)
    _, _, _, w3, ev3 = timed_svd(a, ('bench', SIDE))
    print("         the same matrix again (warm): %.2f ms (events %.2f) warm=%d sweeps %d" % (w3, ev3, sw.stats['warm_calls'], sw.stats['warm_sweeps']), flush=True)
if os.environ.get('PROFILE'):
    cur = drift(cur)
    a = array_of(cur)
    sw.PROFILE = True
    npc.SVD_PROFILE = True
    timed_svd(a, ('bench', SIDE))
    print("stage times (ms, with synchronisation):", {k: round(1e3 * v, 2) for k, v in sw.stats.items() if k.startswith('t_')},
          {k: round(1e3 * v, 2) for k, v in getattr(npc, 'svd_profile', {}).items()} if hasattr(npc, 'svd_profile') else '')
