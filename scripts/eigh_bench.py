"""npc.eigh of a block-diagonal PSD matrix shaped like the mixer's density matrix (blocks ~ (d chi)-sector sizes)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tenpy_amd import _lib
from tenpy_amd.linalg import np_conserved as npc
from tenpy_amd.linalg.charges import ChargeInfo, LegCharge

sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '584,460,460,216,216,52,52').split(',')]
ch = ChargeInfo([1])
leg = LegCharge.from_qind(ch, np.concatenate([[0], np.cumsum(sizes)]), np.arange(len(sizes)).reshape(-1, 1))
rng = np.random.default_rng(0)
dense = np.zeros((sum(sizes), sum(sizes)))
o = 0
for n in sizes:
    r = max(1, n // 2)
    y = rng.standard_normal((n, r)) * np.logspace(0, -8, r)
    dense[o:o + n, o:o + n] = y @ y.T + 1e-5 * np.eye(n)
    o += n
a = npc.Array.from_ndarray(dense, [leg, leg.conj()])
for alg in [int(x) for x in os.environ.get('ALGS', '0,2').split(',')]:
    _lib.load().tpa_svd_set_algorithm(alg)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        w, v = npc.eigh(a)
        torch.cuda.synchronize()
        dt = time.time() - t0
    wr = np.linalg.eigvalsh(dense)
    print("alg=%d  eigh %.2f ms  max|w - w_ref| = %.2e" % (alg, dt * 1e3, np.abs(np.sort(w) - wr).max()), flush=True)
_lib.load().tpa_svd_set_algorithm(0)

# the mixer's case: rank-deficient graded PSD (theta theta^dagger + small perturbation), eigh vs the rank-revealing SVD route
if os.environ.get('PSD', '1') == '1':
    from tenpy_amd.algorithms.mps_common import DensityMatrixMixer
    dense2 = dense - 1e-5 * np.eye(len(dense))
    a2 = npc.Array.from_ndarray(dense2, [leg, leg.conj()])
    wr = np.linalg.eigvalsh(dense2)
    for via in (False, True):
        mx = DensityMatrixMixer(1e-5, eigh_via_svd=via)
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.time()
            w, v = mx._eigh_psd(a2)
            torch.cuda.synchronize()
            dt = time.time() - t0
        vd = v.to_ndarray()
        res = np.abs(dense2 @ vd - vd * w[None, :]).max()
        print("psd via_svd=%s  %.2f ms  max|w - w_ref| = %.2e  max|A v - v w| = %.2e" %
              (via, dt * 1e3, np.abs(np.sort(w) - wr).max(), res), flush=True)
