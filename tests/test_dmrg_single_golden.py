"""Single-site DMRG (OneSiteH + SubspaceExpansion mixer) vs the reference's ``SingleSiteDMRGEngine`` runs dumped by
tests/golden/make_golden.py:gen_dmrg_single -- energy of every site update, truncation errors, on/off schedule of the
mixer, bond dimensions and (after mixer_cleanup) the Schmidt spectra."""
import numpy as np

from helpers import golden
from tenpy_amd.algorithms.dmrg import SingleSiteDMRGEngine, TwoSiteDMRGEngine
from tenpy_amd.algorithms.mps_common import OneSiteH
from tenpy_amd.linalg import np_conserved as npc
from tenpy_amd.models.spin_chains import spin_half_leg, xxz_chain_mpo
from tenpy_amd.networks.mps import MPS


def _setup(rec):
    L = rec['L']
    H = xxz_chain_mpo(L, rec['Jxx'], rec['Jz'], rec['hz'])
    _, p = spin_half_leg('Sz')
    psi = MPS.from_product_state([p] * L, [1, 0] * (L // 2))
    return L, H, psi


def test_single_site_dmrg(backend):
    for rec in golden('dmrg_single.pkl'):
        L, H, psi = _setup(rec)
        if rec['pre_two_site_sweeps']:
            e2 = TwoSiteDMRGEngine(psi, H, {'trunc_params': {'chi_max': rec['chi'], 'svd_min': 1.e-10}, 'lanczos_params': {}})
            for s in range(rec['pre_two_site_sweeps']):
                e2.sweep()
                assert abs(e2.sweep_stats['E'][-1] - rec['E_pre'][s]) <= 1e-10 * abs(rec['E_pre'][s])
        opts = {'trunc_params': {'chi_max': rec['chi'], 'svd_min': rec['svd_min']}, 'lanczos_params': {}}
        if rec['amplitude']:
            opts.update(mixer=True, mixer_params={'amplitude': rec['amplitude'], 'decay': rec['decay'],
                                                  'disable_after': rec['disable_after']})
        else:
            opts.update(mixer=None)
        eng = SingleSiteDMRGEngine(psi, H, opts)
        eng.mixer_activate()
        for s in range(rec['n_sweeps']):
            assert (eng.mixer is not None) == rec['mixer_on'][s]
            eng.sweep()
            assert abs(eng.sweep_stats['E'][-1] - rec['E_sweeps'][s]) <= 1e-10 * abs(rec['E_sweeps'][s])
        assert eng.update_stats['i0'] == rec['i0_updates']
        tol_u = 1e-10 if backend == 'mock' else 1e-8      # (GPU: Jacobi instead of LAPACK rounding in every SVD)
        np.testing.assert_allclose(eng.update_stats['E_total'], rec['E_updates'], rtol=tol_u, atol=tol_u)
        np.testing.assert_allclose(eng.update_stats['err'], rec['err_updates'], rtol=0, atol=1e-11 if backend == 'mock' else 1e-9)
        eng.mixer_cleanup()
        assert list(psi.chi) == rec['chi_final']
        for i in range(1, L):
            S = psi.get_SL(i)
            assert isinstance(S, np.ndarray) and S.ndim == 1
            np.testing.assert_allclose(np.sort(S)[::-1], np.sort(rec['S'][i - 1])[::-1], rtol=0, atol=1e-9)
        np.testing.assert_allclose(psi.entanglement_entropy(), rec['S_ent'], rtol=0, atol=1e-8)
        assert abs(psi.norm_test() - 1.) < 1e-10


def test_one_site_h_matvec_dense(backend):
    """OneSiteH.matvec in both directions == the dense contraction LP-W0-RP applied to theta (numpy einsum)."""
    rec = golden('dmrg_single.pkl')[2]
    L, H, psi = _setup(rec)
    e2 = TwoSiteDMRGEngine(psi, H, {'trunc_params': {'chi_max': 10, 'svd_min': 1.e-10}, 'lanczos_params': {}})
    e2.sweep()
    i0 = L // 2
    LP = e2.env.get_LP(i0).transpose(['vR*', 'wR', 'vR']).to_ndarray()
    RP = e2.env.get_RP(i0).transpose(['vL*', 'wL', 'vL']).to_ndarray()
    W = H.get_W(i0).transpose(['wL', 'wR', 'p', 'p*']).to_ndarray()
    th = psi.get_theta(i0, n=1)
    want = np.einsum('awb,wxpq,cxd,bqd->apc', LP, W, RP, th.transpose(['vL', 'p0', 'vR']).to_ndarray())
    for move_right in (True, False):
        eff = OneSiteH(e2.env, i0, move_right=move_right)
        x = eff.combine_theta(th)
        y = eff.matvec(x)
        assert list(y.get_leg_labels()) == eff.acts_on
        got = y.split_legs().transpose(['vL', 'p0', 'vR']).to_ndarray()
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-12 * max(1., np.abs(want).max()))
        M = eff.to_matrix()
        np.testing.assert_allclose(M @ x.to_ndarray().reshape(-1), y.to_ndarray().reshape(-1), rtol=0, atol=1e-12)
        np.testing.assert_allclose(M, M.conj().T, rtol=0, atol=1e-12)


def test_take_slice(backend):
    rng = np.random.RandomState(5)
    from tenpy_amd.linalg.charges import ChargeInfo, LegCharge
    ch = ChargeInfo([1, 3])
    legs = []
    for n, qc in ((5, 1), (4, -1), (6, 1), (3, -1)):
        q = np.stack([rng.randint(-1, 2, size=n), rng.randint(0, 3, size=n)], axis=1)
        legs.append(LegCharge.from_qflat(ch, q, qc).bunch()[1])
    dense = rng.standard_normal([l.ind_len for l in legs])
    # zero out what charge conservation forbids for qtotal = (0, 1)
    a = npc.Array.from_func(lambda shape: rng.standard_normal(shape), legs, qtotal=[0, 1]) if hasattr(npc.Array, 'from_func') \
        else None
    if a is None:
        import itertools
        qt = np.array([0, 1])
        for idx in itertools.product(*[range(l.ind_len) for l in legs]):
            tot = sum(l.get_charge(l.get_qindex(i)[0]) for l, i in zip(legs, idx))
            if np.any(ch.make_valid(tot) != ch.make_valid(qt)):
                dense[idx] = 0.
        a = npc.Array.from_ndarray(dense, legs, qtotal=qt)
    a.iset_leg_labels(['a', 'b', 'c', 'd'])
    full = a.to_ndarray()
    for axes, idx in ((['b'], [2]), (['a', 'd'], [4, 0]), ([2], [-1]), (['d', 'b', 'a'], [1, 3, 0])):
        r = a.take_slice(idx, axes)
        r.test_sanity()
        sl = [slice(None)] * 4
        for ax, i in zip(a.get_leg_indices(axes), idx):
            sl[ax] = i
        np.testing.assert_array_equal(r.to_ndarray(), full[tuple(sl)])
        assert r.get_leg_labels() == [l for l in ['a', 'b', 'c', 'd'] if a.get_leg_index(l) not in a.get_leg_indices(axes)]
    np.testing.assert_array_equal(a.to_ndarray(), full)      # operand untouched


def test_two_site_dmrg_with_subspace_expansion(backend):
    """Two-site engine with mixer='SubspaceExpansion' (reference ``Mixer.mix_and_decompose_2site`` fallback)."""
    for rec in golden('dmrg_two_site_subspace.pkl'):
        L, H, psi = _setup(rec)
        eng = TwoSiteDMRGEngine(psi, H, {'mixer': 'SubspaceExpansion', 'mixer_params': {'amplitude': rec['amplitude'], 'decay': rec['decay'],
                                                                                         'disable_after': rec['disable_after']},
                                         'trunc_params': {'chi_max': rec['chi'], 'svd_min': rec['svd_min']}, 'lanczos_params': {}})
        eng.mixer_activate()
        for s in range(rec['n_sweeps']):
            assert (eng.mixer is not None) == rec['mixer_on'][s]
            eng.sweep()
            assert abs(eng.sweep_stats['E'][-1] - rec['E_sweeps'][s]) <= 1e-10 * abs(rec['E_sweeps'][s])
        tol_u = 1e-10 if backend == 'mock' else 1e-8
        np.testing.assert_allclose(eng.update_stats['E_total'], rec['E_updates'], rtol=tol_u, atol=tol_u)
        np.testing.assert_allclose(eng.update_stats['err'], rec['err_updates'], rtol=0, atol=1e-11 if backend == 'mock' else 1e-9)
        eng.mixer_cleanup()
        for i in range(1, L):
            np.testing.assert_allclose(np.sort(psi.get_SL(i))[::-1], np.sort(rec['S'][i - 1])[::-1], rtol=0, atol=1e-9)
