"""Mid-size end-to-end parity with runs of the REFERENCE (tests/golden/make_golden.py:gen_midsize): the
stand-alone driver on both backends against TeNPy's TwoSiteDMRGEngine / TEBDEngine -- XXZ L=32 chi=128, Hubbard ladder 2x6
chi=128 (U(1)xU(1)), TFI-parity real-time TEBD L=16 chi=64 (complex128)."""
import numpy as np
import pytest

from helpers import golden
from tenpy_amd.algorithms.dmrg import TwoSiteDMRGEngine
from tenpy_amd.algorithms.tebd import TEBDEngine
from tenpy_amd.models.hubbard import hubbard_ladder_mpo, spinful_fermion_leg
from tenpy_amd.models.spin_chains import spin_half_leg, xxz_chain_mpo
from tenpy_amd.networks.mps import MPS


def test_xxz_L32_chi128(backend):
    rec = golden('midsize.pkl')['xxz_L32_chi128']
    L = rec['L']
    H = xxz_chain_mpo(L, 1., 1., 0.)
    _, p = spin_half_leg('Sz')
    psi = MPS.from_product_state([p] * L, [1, 0] * (L // 2))
    eng = TwoSiteDMRGEngine(psi, H, {'trunc_params': {'chi_max': rec['chi'], 'svd_min': 1.e-10}})
    for s, E in enumerate(rec['E_sweeps']):
        eng.sweep()
        # un-converged sweeps amplify rounding differences (which noise-level Schmidt values survive svd_min); converged: 1e-10
        tol = 1e-7 if s < 2 else 1e-10
        assert abs(eng.sweep_stats['E'][-1] - E) <= tol * abs(E), (s, eng.sweep_stats['E'][-1], E)
    assert max(psi.chi) == rec['chi_final']
    S, Sref = np.sort(psi.get_SL(L // 2))[::-1], np.sort(rec['S_mid'])[::-1]
    np.testing.assert_allclose(S, Sref, rtol=0, atol=1e-9 * Sref[0])
    np.testing.assert_allclose(psi.entanglement_entropy(), rec['S_ent'], rtol=0, atol=1e-8)


def test_hubbard_2x6_chi128(backend):
    rec = golden('midsize.pkl')['hubbard_2x6_chi128']
    Lx = rec['Lx']
    H = hubbard_ladder_mpo(Lx, rec['t'], rec['U'], rec['mu'])
    _, p = spinful_fermion_leg()
    psi = MPS.from_product_state([p] * (2 * Lx), [1, 2] * Lx)
    eng = TwoSiteDMRGEngine(psi, H, {'trunc_params': {'chi_max': rec['chi'], 'svd_min': 1.e-10}})
    for s, E in enumerate(rec['E_sweeps']):
        eng.sweep()
        tol = 1e-5 if s < 3 else 1e-9          # (the two MPOs differ by a gauge of the virtual index: different Lanczos paths)
        assert abs(eng.sweep_stats['E'][-1] - E) <= tol * abs(E), (s, eng.sweep_stats['E'][-1], E)
    assert max(psi.chi) == rec['chi_final']
    np.testing.assert_allclose(psi.entanglement_entropy(), rec['S_ent'], rtol=0, atol=1e-6)


def test_tebd_tfi_parity_L16_chi64(backend):
    rec = golden('midsize.pkl')['tebd_tfi_parity_L16_chi64']
    L = rec['L']
    _, p = spin_half_leg('parity')
    up = dict(rec['state_labels'])['up']
    psi = MPS.from_product_state([p] * L, [up] * L, dtype=np.complex128)
    eng = TEBDEngine(psi, rec['h_bond'], {'dt': rec['dt'], 'trunc_params': {'chi_max': rec['chi'], 'svd_min': 1.e-10}})
    k = 0
    for step in range(rec['every'] * len(rec['chi_t'])):
        eng.evolve_step_order2()
        if step % rec['every'] == rec['every'] - 1:
            assert max(psi.chi) == rec['chi_t'][k]
            np.testing.assert_allclose(psi.entanglement_entropy(), rec['S_t'][k], rtol=0, atol=1e-9)
            k += 1
    np.testing.assert_allclose(np.sort(psi.get_SL(L // 2))[::-1], np.sort(rec['S_mid'])[::-1], rtol=0, atol=1e-9)
    assert psi.get_B(0, None).dtype == np.complex128
