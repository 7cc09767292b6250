"""Run the two-site DMRG harness on the GPU: python scripts/run_dmrg.py L chi n_sweeps [model]"""
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tenpy_amd.models.spin_chains import xxz_chain_mpo, tfi_chain_mpo, spin_half_leg
from tenpy_amd.networks.mps import MPS
from tenpy_amd.algorithms.dmrg import TwoSiteDMRGEngine

from tenpy_amd import _lib
if os.environ.get('SVD_ALG'):
    _lib.load().tpa_svd_set_algorithm(int(os.environ['SVD_ALG']))
L = int(sys.argv[1]) if len(sys.argv) > 1 else 16
chi = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 4
model = sys.argv[4] if len(sys.argv) > 4 else 'xxz'
if model == 'xxz':
    H = xxz_chain_mpo(L, 1., 1., 0.)
    chinfo, p = spin_half_leg('Sz')
    psi = MPS.from_product_state([p] * L, [1, 0] * (L // 2))
elif model == 'hubbard':
    from tenpy_amd.models.hubbard import hubbard_ladder_mpo, spinful_fermion_leg
    H = hubbard_ladder_mpo(L // 2, 1., 8., 0.)
    chinfo, p = spinful_fermion_leg()
    psi = MPS.from_product_state([p] * L, [1, 2] * (L // 2))
else:
    H = tfi_chain_mpo(L, 1., 1., None)
    chinfo, p = spin_half_leg(None)
    psi = MPS.from_product_state([p] * L, [1] * L)
chi_list = {0: 64}
_c, _s = 64, 2
while _c < chi:
    _c = min(2 * _c, chi)
    chi_list[_s] = _c
    _s += 1
print('chi_list', chi_list)
eng = TwoSiteDMRGEngine(psi, H, {'chi_list': chi_list, 'trunc_params': {'chi_max': chi, 'svd_min': float(os.environ.get('SVD_MIN', 1e-10))}, 'lanczos_params': ({'N_min': int(os.environ['NLANCZOS']), 'N_max': int(os.environ['NLANCZOS'])} if os.environ.get('NLANCZOS') else {}),
                                 'profile': True, 'mixer': bool(os.environ.get('MIXER')), 'mixer_params': {'amplitude': 1.e-5, 'decay': 2., 'disable_after': 15}})
if os.environ.get('MIXER'):
    eng.mixer_activate()
for s in range(ns):
    torch.cuda.synchronize()
    t = time.time()
    eng.sweep()
    torch.cuda.synchronize()
    print("sweep %d E=%.13f chi=%d t=%.3fs" % (s, eng.sweep_stats['E'][-1], eng.sweep_stats['max_chi'][-1], time.time() - t), {k: round(v, 3) for k, v in eng.phase_time.items()}, 'N_lanczos', int(np.sum(eng.update_stats['N_lanczos'][-2*(L-2):])), flush=True)
    eng.phase_time = {k: 0. for k in eng.phase_time}
    from tenpy_amd.linalg import np_conserved as _npc
    print('   svd stats', _npc.svd_stats, flush=True)
    _npc.svd_stats.update(calls=0, sweeps=0, max_block=0)
