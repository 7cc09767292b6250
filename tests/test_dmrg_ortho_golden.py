"""Excited states: DMRG with ``orthogonal_to=[ground state]`` (``OrthogonalNpcLinearOperator`` around the effective
Hamiltonian, ``MPSEnvironment`` overlaps) vs the reference's runs (tests/golden/make_golden.py:gen_dmrg_ortho)."""
import numpy as np

from helpers import golden
from tenpy_amd.algorithms.dmrg import SingleSiteDMRGEngine, TwoSiteDMRGEngine
from tenpy_amd.models.spin_chains import spin_half_leg, xxz_chain_mpo
from tenpy_amd.networks.mps import MPS, MPSEnvironment


def test_excited_state_dmrg(backend):
    recs = golden('dmrg_ortho.pkl')
    L = recs[0]['L']
    H = xxz_chain_mpo(L, recs[0]['Jxx'], recs[0]['Jz'], recs[0]['hz'])
    _, p = spin_half_leg('Sz')
    psi0 = MPS.from_product_state([p] * L, [1, 0] * (L // 2))
    e0 = TwoSiteDMRGEngine(psi0, H, {'trunc_params': {'chi_max': recs[0]['chi'], 'svd_min': 1.e-10}, 'lanczos_params': {}})
    for s, E in enumerate(recs[0]['E0_sweeps']):
        e0.sweep()
        assert abs(e0.sweep_stats['E'][-1] - E) <= 1e-10 * abs(E)
    for rec in recs:
        psi1 = MPS.from_product_state([p] * L, [0, 1] * (L // 2))
        opts = {'trunc_params': {'chi_max': rec['chi'], 'svd_min': rec['svd_min']}, 'lanczos_params': {}}
        if rec['engine'] == 'single':
            opts.update(mixer=True, mixer_params={'amplitude': 1.e-3, 'decay': 2., 'disable_after': 3})
            e1 = SingleSiteDMRGEngine(psi1, H, opts, orthogonal_to=[psi0])
        else:
            e1 = TwoSiteDMRGEngine(psi1, H, opts, orthogonal_to=[psi0])
        e1.mixer_activate()
        for s, E in enumerate(rec['E1_sweeps']):
            e1.sweep()
            assert abs(e1.sweep_stats['E'][-1] - E) <= 1e-9 * abs(E), (rec['engine'], s)
        np.testing.assert_allclose(e1.update_stats['E_total'], rec['E1_updates'], rtol=1e-8, atol=1e-8)
        e1.mixer_cleanup()
        ov = MPSEnvironment(psi0, psi1).full_contraction(L // 2 - 1)
        assert abs(ov) < 1e-9
        assert abs(MPSEnvironment(psi1, psi1).full_contraction(L // 2 - 1) - 1.) < 1e-10
        assert e1.sweep_stats['E'][-1] > e0.sweep_stats['E'][-1] + 0.1


def test_dmrg_with_exact_diagonalisation_of_small_bonds(backend):
    """``diag_method='default'`` (the reference's default: ED below max_N_for_ED = 400, Lanczos above) and 'ED_block'
    (tests/golden/make_golden.py:gen_dmrg_default_diag): same energies update by update, same choice ED / Lanczos."""
    for rec in golden('dmrg_default_diag.pkl'):
        L = rec['L']
        H = xxz_chain_mpo(L, rec['Jxx'], rec['Jz'], rec['hz'])
        _, p = spin_half_leg('Sz')
        psi = MPS.from_product_state([p] * L, [1, 0] * (L // 2))
        opts = {'diag_method': rec['diag_method'], 'trunc_params': {'chi_max': rec['chi'], 'svd_min': rec['svd_min']}, 'lanczos_params': {}}
        if rec['engine'] == 'two':
            eng = TwoSiteDMRGEngine(psi, H, opts)
        else:
            opts.update(mixer=True, mixer_params={'amplitude': 1.e-3, 'decay': 2., 'disable_after': 3})
            eng = SingleSiteDMRGEngine(psi, H, opts)
        eng.mixer_activate()
        for s, E in enumerate(rec['E_sweeps']):
            eng.sweep()
            assert abs(eng.sweep_stats['E'][-1] - E) <= 1e-10 * abs(E)
        np.testing.assert_allclose(eng.update_stats['E_total'], rec['E_updates'], rtol=1e-10, atol=1e-10)
        assert [n == -1 for n in eng.update_stats['N_lanczos']] == [n == -1 for n in rec['N_lanczos']]
        eng.mixer_cleanup()
        for i in range(1, L):
            np.testing.assert_allclose(np.sort(psi.get_SL(i))[::-1], np.sort(rec['S'][i - 1])[::-1], rtol=0, atol=1e-9)
