"""Infinite DMRG (two-site unit cell, environments growing across the unit-cell boundary, energy per site from the
ages of the environment parts) vs the reference's run (tests/golden/make_golden.py:gen_idmrg).  The TFI value is the
golden number of the reference's own test (tests/test_dmrg.py:121: -1.67192622)."""
import numpy as np

from helpers import golden
from tenpy_amd.algorithms.dmrg import TwoSiteDMRGEngine
from tenpy_amd.models.spin_chains import spin_half_leg, tfi_chain_mpo, xxz_chain_mpo
from tenpy_amd.networks.mps import MPS


def test_idmrg(backend):
    for rec in golden('idmrg.pkl'):
        L = rec['L']
        if rec['case'].startswith('xxz'):
            H = xxz_chain_mpo(L, rec['Jxx'], rec['Jz'], rec['hz'], bc='infinite')
            _, p = spin_half_leg('Sz')
            psi = MPS.from_product_state([p] * L, [1, 0], bc='infinite')
        else:
            H = tfi_chain_mpo(L, rec['J'], rec['g'], conserve='parity', bc='infinite')
            _, p = spin_half_leg('parity')
            psi = MPS.from_product_state([p] * L, [1, 1], bc='infinite')
        opts = {k: v for k, v in rec['options'].items() if k not in ('combine', 'max_N_for_ED')}
        eng = TwoSiteDMRGEngine(psi, H, opts)
        E, _ = eng.run()
        assert eng.sweeps == rec['sweeps']
        assert abs(E - rec['E']) < 1e-10
        if rec['case'] == 'tfi':
            assert abs(E - (-1.67192622)) < 1e-6
        assert eng.update_stats['i0'] == rec['i0']
        assert eng.update_stats['age'] == rec['age']
        # (the first environments are the dominant eigenvectors of the MPO transfer matrix, like the reference's)
        tol_u = 1e-10 if backend == 'mock' else 1e-8
        np.testing.assert_allclose(eng.update_stats['E_total'], rec['E_updates'], rtol=tol_u, atol=10 * tol_u)
        for k, tol in (('sweep', 0), ('N_updates', 0), ('E', 1e-10), ('Delta_E', 1e-10), ('S', 1e-8), ('Delta_S', 1e-8), ('max_S', 1e-8),
                       ('max_trunc_err', 1e-11), ('max_E_trunc', 1e-9), ('max_chi', 0)):
            a, b = np.array(eng.sweep_stats[k], dtype=float), np.array(rec['sweep_stats'][k], dtype=float)
            assert a.shape == b.shape, k
            if backend != 'mock' and tol > 0:
                tol = max(100 * tol, 1e-8)
            assert np.all(np.isnan(b) | (np.abs(a - b) <= tol * np.maximum(1., np.abs(b)))), (k, a, b)
        assert list(psi.chi) == rec['chi']
        for i in range(L):
            # (run() ends like the reference's: environment sweeps / psi.canonical_form() until the norm error is < 1e-10)
            np.testing.assert_allclose(np.sort(psi.get_SL(i))[::-1], np.sort(rec['S'][i])[::-1], rtol=0, atol=1e-8)


def test_idmrg_benchmark_model(backend):
    """The reference's iDMRG benchmark in small (tests/benchmark/dmrg_infinite.py: spin-2 chain, D = 0.3, Sz conserved,
    Lanczos N_min = N_max = 10, optimisation sweeps alternating with environment sweeps)."""
    from tenpy_amd.models.spin_chains import spin_S_leg, spin_chain_mpo
    rec = golden('idmrg_bench.pkl')[0]
    L = rec['L']
    H = spin_chain_mpo(L, S=rec['S'], D=rec['D'], bc='infinite')
    _, p = spin_S_leg(rec['S'])
    labels = dict(rec['state_labels'])
    d = p.ind_len
    # this package orders the states m = -S ... S; 'up' = m = +S, 'down' = m = -S
    idx = {'up': d - 1, 'down': 0}
    psi = MPS.from_product_state([p] * L, [idx[s] for s in (['up', 'down'] * L)[:L]], bc='infinite')
    eng = TwoSiteDMRGEngine(psi, H, {'trunc_params': {'chi_max': rec['chi'], 'svd_min': 1.e-10}, 'lanczos_params': {'N_min': 10, 'N_max': 10}})
    n0 = len(eng.update_stats['E_total'])        # (the engine starts with one environment sweep, like the reference)
    for i in range(6):
        eng.sweep()
        eng.sweep(optimize=False)
    assert eng.update_stats['i0'] == rec['i0']
    assert eng.update_stats['age'] == rec['age']
    np.testing.assert_allclose(eng.update_stats['E_total'], rec['E_updates'], rtol=1e-10 if backend == 'mock' else 1e-8, atol=1e-8 if backend == 'mock' else 1e-6)
    assert list(psi.chi) == rec['chi_final']
    np.testing.assert_allclose(psi.entanglement_entropy(), rec['S_ent'], rtol=0, atol=1e-7)
    assert labels['up'] in (0, d - 1) and n0 > 0


def test_idmrg_single_site(backend):
    """Single-site infinite DMRG with the subspace expansion vs the reference's run."""
    from tenpy_amd.algorithms.dmrg import SingleSiteDMRGEngine
    rec = golden('idmrg_single.pkl')[0]
    L = rec['L']
    H = xxz_chain_mpo(L, rec['Jxx'], rec['Jz'], rec['hz'], bc='infinite')
    _, p = spin_half_leg('Sz')
    psi = MPS.from_product_state([p] * L, [1, 0], bc='infinite')
    opts = {k: v for k, v in rec['options'].items() if k not in ('combine', 'max_N_for_ED')}
    eng = SingleSiteDMRGEngine(psi, H, opts)
    E, _ = eng.run()
    assert eng.sweeps == rec['sweeps'] and abs(E - rec['E']) < 1e-9
    assert eng.update_stats['i0'] == rec['i0'] and eng.update_stats['age'] == rec['age']
    tol_u = 1e-9 if backend == 'mock' else 1e-7
    np.testing.assert_allclose(eng.update_stats['E_total'], rec['E_updates'], rtol=tol_u, atol=10 * tol_u)
    for k in ('E', 'Delta_E', 'S'):
        a, b = np.array(eng.sweep_stats[k], dtype=float), np.array(rec['sweep_stats'][k], dtype=float)
        assert a.shape == b.shape and np.all(np.isnan(b) | (np.abs(a - b) < 1e-7)), (k, a, b)
    assert list(psi.chi) == rec['chi']
    for i in range(L):
        np.testing.assert_allclose(np.sort(psi.get_SL(i))[::-1], np.sort(rec['S'][i])[::-1], rtol=0, atol=1e-7)
