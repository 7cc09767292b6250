"""Host wall time of the segments of a bond update of the stand-alone driver (no added synchronisation: segments that only enqueue
work show their pure host cost, segments that wait for the device show the wait): python scripts/host_segments.py L chi n_sweeps"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
import numpy as np
import torch
from tenpy_amd.models.spin_chains import xxz_chain_mpo, spin_half_leg
from tenpy_amd.networks.mps import MPS
from tenpy_amd.algorithms import dmrg as dm
from tenpy_amd.linalg import np_conserved as npc
from tenpy_amd.linalg import truncation

L, chi, ns = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
seg = collections.defaultdict(float)


def update_bond(self, i0, move_right=True):
    T = time.perf_counter
    t = [T()]

    def mark(name):
        now = T()
        seg[name] += now - t[0]
        t[0] = now
    psi, env = self.psi, self.env
    eff_H = dm.TwoSiteH(env, i0, combine=True, move_right=move_right)
    mark('A TwoSiteH()')
    theta = eff_H.combine_theta(psi.get_theta(i0, n=2))
    mark('B get_theta + combine_theta')
    E0, theta, N = dm.LanczosGroundState(eff_H, theta, self.lanczos_params).run()
    mark('C Lanczos.run (waits)')
    theta = eff_H.prepare_svd(theta)
    mark('D prepare_svd')
    npc.svd_hint = ((id(self), i0), 'R' if move_right else 'L')
    try:
        U, S, VH, err, _ = dm.svd_theta(theta, self.trunc_params, qtotal_LR=[psi.get_B(i0, None).qtotal, None], inner_labels=['vR', 'vL'])
    finally:
        npc.svd_hint = None
    mark('E svd_theta (waits)')
    i1 = i0 + 1
    if move_right:
        eff_H.update_LP(env, i1, U)
    else:
        eff_H.update_RP(env, i0, VH)
    mark('F update_LP/RP')
    psi.set_B(i0, U.split_legs(['(vL.p0)']).ireplace_label('p0', 'p'), form='A')
    psi.set_B(i1, VH.split_legs(['(p1.vR)']).ireplace_label('p1', 'p'), form='B')
    psi.set_SR(i0, S)
    env.invalidate(i0, i1, keep_LP=move_right, keep_RP=not move_right)
    mark('G set_B + invalidate')
    us = self.update_stats
    us['i0'].append(i0)
    us['E_total'].append(float(E0))
    us['N_lanczos'].append(N)
    us['time'].append(0.)
    us['err'].append(err.eps)
    us['chi'].append(len(S))
    mark('H stats')
    return err


# finer split of svd_theta
orig_svd = npc.svd
orig_trunc = truncation.truncate
sub = collections.defaultdict(float)


def svd_timed(*a, **k):
    t0 = time.perf_counter()
    r = orig_svd(*a, **k)
    sub['npc.svd'] += time.perf_counter() - t0
    return r


npc.svd = svd_timed
truncation.svd = svd_timed if hasattr(truncation, 'svd') else None
H = xxz_chain_mpo(L, 1., 1., 0.)
chinfo, p = spin_half_leg('Sz')
psi = MPS.from_product_state([p] * L, [1, 0] * (L // 2))
chi_list, c, s = {0: 64}, 64, 2
while c < chi:
    c = min(2 * c, chi)
    chi_list[s] = c
    s += 1
eng = dm.TwoSiteDMRGEngine(psi, H, {'chi_list': chi_list, 'trunc_params': {'chi_max': chi, 'svd_min': 1e-14},
                                    'lanczos_params': {'N_min': 8, 'N_max': 8}})
dm.TwoSiteDMRGEngine.update_bond = update_bond
for sw in range(len(chi_list) + 1 + ns):
    seg.clear()
    sub.clear()
    torch.cuda.synchronize()
    t0 = time.time()
    eng.sweep()
    torch.cuda.synchronize()
    dt = time.time() - t0
    nb = 2 * (L - 2)
    print('sweep %d chi %d: %.3f s; ms/bond: %s | npc.svd %.2f' % (sw, eng.sweep_stats['max_chi'][-1], dt,
          {k: round(1e3 * v / nb, 2) for k, v in sorted(seg.items())}, 1e3 * sub['npc.svd'] / nb), flush=True)
