"""Where does the time of one svd_theta go at chi=2048?  Grows the state like bench.py, then times on the centre bond:
the bare C call tpa_svd_batch, npc.svd (Python wrapper around it) and truncation.svd_theta (adds truncation + projection)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tenpy_amd.algorithms.dmrg import TwoSiteDMRGEngine
from tenpy_amd.algorithms.mps_common import TwoSiteH
from tenpy_amd.linalg import np_conserved as npc
from tenpy_amd.linalg.truncation import svd_theta
from tenpy_amd.models.spin_chains import spin_half_leg, xxz_chain_mpo
from tenpy_amd.networks.mps import MPS

L, chi = 100, int(sys.argv[1]) if len(sys.argv) > 1 else 2048
H = xxz_chain_mpo(L, 1., 1., 0.)
_, p = spin_half_leg('Sz')
psi = MPS.from_product_state([p] * L, [1, 0] * (L // 2))
eng = TwoSiteDMRGEngine(psi, H, {'trunc_params': {'chi_max': 64, 'svd_min': 1.e-14}, 'lanczos_params': {'N_min': 2, 'N_max': 20}})
c = 64
eng.sweep(); eng.sweep()
while c < chi:
    c = min(2 * c, chi)
    eng.trunc_params['chi_max'] = c
    eng.sweep()
for _ in range(int(os.environ.get('EXTRA_SWEEPS', 0))):
    eng.sweep()
i0 = L // 2 - 1
eff = TwoSiteH(eng.env, i0)
theta = eff.prepare_svd(eff.combine_theta(psi.get_theta(i0, n=2)))
print("theta blocks", sorted([tuple(int(x) for x in s) for s in theta._block_shapes()], reverse=True)[:6], flush=True)


def timeit(f, reps=5):
    f()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.time() - t0) / reps * 1e3


tp = {'chi_max': chi, 'svd_min': 1.e-14}
if os.environ.get('SVD_ALG'):       # e.g. SVD_ALG=1024: predicted convergence; compare singular values / orthogonality with the default
    from tenpy_amd import _lib
    U0, S0, V0 = npc.svd(theta, inner_labels=['vR', 'vL'])
    _lib.load().tpa_svd_set_algorithm(int(os.environ['SVD_ALG']))
    npc.svd_stats.update(calls=0, sweeps=0)
    U1, S1, V1 = npc.svd(theta, inner_labels=['vR', 'vL'])
    u = U1.to_ndarray()
    keep = S1 > 1e-6 * S1.max()
    print("SVD_ALG=%s: sweeps %d, max |S - S_default| = %.2e, |U^T U - 1| (sigma > 1e-6) = %.2e, |U S V - theta| = %.2e"
          % (os.environ['SVD_ALG'], npc.svd_stats['sweeps'], np.abs(S1 - S0).max(),
             np.abs(u[:, keep].conj().T @ u[:, keep] - np.eye(keep.sum())).max(),
             np.abs((u * S1) @ V1.to_ndarray() - theta.to_ndarray()).max()), flush=True)
t_theta = timeit(lambda: svd_theta(theta, tp, qtotal_LR=[psi.get_B(i0, None).qtotal, None], inner_labels=['vR', 'vL']))
t_svd = timeit(lambda: npc.svd(theta, inner_labels=['vR', 'vL']))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    svd_theta(theta, tp, qtotal_LR=[psi.get_B(i0, None).qtotal, None], inner_labels=['vR', 'vL'])
torch.cuda.synchronize()
pr.disable()
print("svd_theta %.2f ms   npc.svd %.2f ms   (sweeps per call %.1f)" % (t_theta, t_svd, npc.svd_stats['sweeps'] / max(npc.svd_stats['calls'], 1)))
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
if os.environ.get('DUMP_THETA'):
    os.makedirs(os.path.dirname(os.environ['DUMP_THETA']), exist_ok=True)
    np.savez(os.environ['DUMP_THETA'], *[np.asarray(b) for b in theta._data])
    print("dumped", os.environ['DUMP_THETA'], flush=True)
