"""Dev harness: stage times of the warm-started block SVD (tenpy_amd/linalg/_svd_warm.py) on the dumped chi=2048 theta."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
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
a = npc.Array([legL, legR], np.float64)
a._set_blocks(np.stack([np.arange(len(ms)), np.arange(len(ms))], axis=1), qdata_sorted=True)
a._arena = torch.from_numpy(np.concatenate([b.reshape(-1) for b in blocks])).cuda()
times = {}
sw.DEBUG = bool(os.environ.get('DEBUG'))


def wrap(mod, name):
    f = getattr(mod, name)

    def g(*args, **kw):
        torch.cuda.synchronize()
        t0 = time.time()
        r = f(*args, **kw)
        torch.cuda.synchronize()
        times[name] = times.get(name, 0.) + time.time() - t0
        return r
    setattr(mod, name, g)


if os.environ.get('STAGES'):
    for n in ('raw_gemm', 'raw_copy', '_row_norms_sq', 'lowdin_rows', '_axpy'):
        wrap(sw, n)
    wrap(npc, '_svd_batch_robust')
    wrap(npc, '_svd_clean_small')
    wrap(npc, '_svd_warm_store')
rng = np.random.RandomState(0)
ALWAYS = bool(os.environ.get('ALWAYS_PERTURB'))
base_arena = a._arena.clone()
for rep in range(int(os.environ.get('REPS', 4))):
    if ALWAYS and rep > 0:      # a different full-rank perturbation before every call: every warm attempt finds a stale basis
        a = a.copy()
        a._arena = base_arena * (1. + float(os.environ.get('PERTURB', 1e-9)) * torch.randn_like(base_arena))
        sw.cooldown.clear()
    if rep == 2 and os.environ.get('PERTURB') and not ALWAYS:      # a slightly different matrix in the same row space
        eps = float(os.environ['PERTURB'])
        a2 = a.copy()
        rk = int(os.environ.get('PRANK', 0))
        if rk == 0:
            a2._arena = a._arena * (1. + eps * torch.randn_like(a._arena))
        else:       # low-rank perturbation of every block
            pieces = []
            for b in blocks:
                x = rng.standard_normal((b.shape[0], rk)) @ rng.standard_normal((rk, b.shape[1]))
                pieces.append((b + eps * np.linalg.norm(b) / np.linalg.norm(x) * x).reshape(-1))
            a2._arena = torch.from_numpy(np.concatenate(pieces)).cuda()
        a = a2
    times.clear()
    for k in sw.stats:
        sw.stats[k] = 0
    npc.svd_hint = ('bench', os.environ.get('SIDE', 'R'))
    if os.environ.get('BURST'):      # a burst of MFMA work before the (latency-bound) SVD: does the clock / power state matter?
        n = 4096
        if 'bx' not in globals():
            bx = torch.randn(n * n, dtype=torch.float64, device='cuda')
            by = torch.empty(n * n, dtype=torch.float64, device='cuda')
        for _ in range(int(os.environ['BURST'])):
            sw.raw_gemm(np.float64, np.array([[0, n, n, n, 0, n, 1, 0, n, 1, n, 0]]), bx, bx, by)
    torch.cuda.synchronize()
    t0 = time.time()
    U, S, VH = npc.svd(a)
    torch.cuda.synchronize()
    t = time.time() - t0
    print("call %d: %.2f ms  warm=%d cold=%d sweeps(w/c)=%d/%d stale=%d" % (rep, 1e3 * t, sw.stats['warm_calls'], sw.stats['cold_calls'],
          sw.stats['warm_sweeps'], sw.stats['cold_sweeps'], sw.stats['fb_stale']),
          {k: round(1e3 * v, 2) for k, v in times.items()}, flush=True)
if os.environ.get('CHECK'):
    Ud, Vd = U.to_ndarray(), VH.to_ndarray()
    Ad = a.to_ndarray()
    print("recon", np.abs((Ud * S) @ Vd - Ad).max() / np.abs(Ad).max())
    keep = S > 1e-14 * S.max()
    print("iso", np.abs(Ud[:, keep].T @ Ud[:, keep] - np.eye(keep.sum())).max(), np.abs(Vd[keep] @ Vd[keep].T - np.eye(keep.sum())).max())
