"""Where does the wall time of npc.svd go at a given chi?  Stand-alone driver, TPA_SVD_PROFILE=1 (synchronises around the stages of
every svd()): python scripts/svd_stage_profile.py L chi n_sweeps"""
import os
import sys
import time
os.environ['TPA_SVD_PROFILE'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tenpy_amd.models.spin_chains import xxz_chain_mpo, spin_half_leg
from tenpy_amd.networks.mps import MPS
from tenpy_amd.algorithms.dmrg import TwoSiteDMRGEngine
from tenpy_amd.linalg import np_conserved as npc, _svd_warm, truncation

L, chi, ns = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
H = xxz_chain_mpo(L, 1., 1., 0.)
chinfo, p = spin_half_leg('Sz')
psi = MPS.from_product_state([p] * L, [1, 0] * (L // 2))
chi_list, c, s = {0: 64}, 64, 2
while c < chi:
    c = min(2 * c, chi)
    chi_list[s] = c
    s += 1
eng = TwoSiteDMRGEngine(psi, H, {'chi_list': chi_list, 'trunc_params': {'chi_max': chi, 'svd_min': 1e-10},
                                 'lanczos_params': {'N_min': 8, 'N_max': 8}, 'profile': True})
orig = truncation.svd_theta
import tenpy_amd.algorithms.dmrg as dm
tt = [0.]


def timed(*a, **k):
    torch.cuda.synchronize()
    t = time.time()
    r = orig(*a, **k)
    torch.cuda.synchronize()
    tt[0] += time.time() - t
    return r


dm.svd_theta = timed
for sw in range(len(chi_list) + 1 + ns):
    for k in list(_svd_warm.stats):
        if k.startswith('t_'):
            _svd_warm.stats[k] = 0.
    tt[0] = 0.
    eng.phase_time = {k: 0. for k in eng.phase_time}
    w0, c0 = _svd_warm.stats['warm_calls'], _svd_warm.stats['cold_calls']
    t = time.time()
    eng.sweep()
    torch.cuda.synchronize()
    nb = 2 * (L - 2)
    print('sweep %d chi %d  %.3f s  phases(ms/bond) %s  svd_theta %.2f ms/bond  stages(ms/bond) %s  warm/cold %d/%d' % (
        sw, eng.sweep_stats['max_chi'][-1], time.time() - t, {k: round(1e3 * v / nb, 2) for k, v in eng.phase_time.items()},
        1e3 * tt[0] / nb, {k: round(1e3 * v / nb, 2) for k, v in _svd_warm.stats.items() if k.startswith('t_')},
        _svd_warm.stats['warm_calls'] - w0, _svd_warm.stats['cold_calls'] - c0), flush=True)
