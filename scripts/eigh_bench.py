"""npc.eigh of a block-diagonal real matrix shaped like the mixer's density matrix (graded, rank-deficient PSD blocks; `mix_rho`, reference
mps_common.py:1972-2079): the checked SVD route (np_conserved.EIGH_VIA_SVD, default for real data) against the two-sided Jacobi iteration of
tpa_eigh_batch.  python scripts/eigh_bench.py [sizes] [graded|graded14|flat|diag]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tenpy_amd.linalg import np_conserved as npc
from tenpy_amd.linalg.charges import ChargeInfo, LegCharge

sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '1086,871,871,450,450,148,148').split(',')]
spec = sys.argv[2] if len(sys.argv) > 2 else 'graded'
ch = ChargeInfo([1])
leg = LegCharge.from_qind(ch, np.concatenate([[0], np.cumsum(sizes)]), np.arange(len(sizes)).reshape(-1, 1))
rng = np.random.default_rng(0)
dense = np.zeros((sum(sizes), sum(sizes)))
o = 0
for n in sizes:
    r = max(1, n // 2)
    if spec == 'graded':
        y = rng.standard_normal((n, r)) * np.logspace(0, -8, r)
        h = y @ y.T
    elif spec == 'graded14':
        q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        h = (q * np.logspace(0, -14, n)) @ q.T
    elif spec == 'diag':          # nearly diagonal in the given basis: diag(S^2) + a 1e-5 perturbation of rank n / 4 (a mixer step)
        z = rng.standard_normal((n, n // 4)) * 1e-3
        h = np.diag(np.logspace(0, -20, n)) + 1e-5 * (z @ z.T)
    else:
        x = rng.standard_normal((n, n))
        h = x.T @ x / n
    dense[o:o + n, o:o + n] = 0.5 * (h + h.T)
    o += n
a = npc.Array.from_ndarray(dense, [leg, leg.conj()])
wr = np.linalg.eigvalsh(dense)
nrm = np.linalg.norm(dense, 2)
for via in (True, False, True):
    npc.EIGH_VIA_SVD = via
    t = 1e9
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.time()
        w, v = npc.eigh(a)
        torch.cuda.synchronize()
        t = min(t, time.time() - t0)
    vd = v.to_ndarray()
    print("%s sizes %s via_svd=%s: %.2f ms  |w - w_lapack|/|A| %.1e  |A v - v w|/|A| %.1e  |v^T v - 1| %.1e  stats %s"
          % (spec, sizes, via, t * 1e3, np.abs(np.sort(w) - wr).max() / nrm, np.abs(dense @ vd - vd * w[None, :]).max() / nrm,
             np.abs(vd.T @ vd - np.eye(len(vd))).max(), npc.eigh_stats), flush=True)
