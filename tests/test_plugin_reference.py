"""The plugin of SURVEY 8(b) against the REAL reference: an unmodified TeNPy two-site DMRG run with
``tenpy_amd.plugin.install`` (Lanczos + block SVD through our boundary, here on the numpy emulation of the device
entry points) must reproduce the same run without the plugin, bond update by bond update.

CPU container only: needs the reference tree at /root/reference (absent on the GPU box -> skipped there)."""
import os
import sys
import warnings

import numpy as np
import pytest

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'tenpy')), reason="reference tree not available")


def _run_reference(tenpy, L, chi, n_sweeps, mixer=None):
    from tenpy.algorithms import dmrg
    from tenpy.models.xxz_chain import XXZChain
    from tenpy.networks.mps import MPS
    M = XXZChain({'L': L, 'Jxx': 1., 'Jz': 1.2, 'hz': 0.1, 'bc_MPS': 'finite', 'sort_charge': True})
    psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * (L // 2), bc='finite')
    eng = dmrg.TwoSiteDMRGEngine(psi, M, {'mixer': mixer, 'combine': True, 'max_N_for_ED': 0,
                                          'trunc_params': {'chi_max': chi, 'svd_min': 1.e-6 if mixer else 1.e-10}})   # (mixer: see gen_dmrg_mixer)
    eng.mixer_activate()
    for _ in range(n_sweeps):
        eng.sweep()
    eng.mixer_cleanup()
    return (np.array(eng.update_stats['E_total']), np.array([e.eps for e in eng.update_stats['err']]),
            [np.array(psi.get_SL(i)) for i in range(1, L)], list(eng.update_stats['N_lanczos']))


@pytest.mark.parametrize("factored", [False, True])
def test_plugin_reproduces_reference_dmrg(backend, factored, monkeypatch):
    """factored: the device twin applies LP . theta . (W0 W1) . RP instead of LHeff . theta . RHeff (forced on here; by
    default it is chosen for bond sectors >= 200 wide) -- the reference's own run is the judge for both forms."""
    if backend != 'mock':
        pytest.skip("reference tree is not on the GPU box")
    from tenpy_amd.algorithms import mps_common
    monkeypatch.setattr(mps_common, 'FACTORED_MIN_SECTOR', 0 if factored else 10**9)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        import tenpy
        from tenpy_amd import plugin
        E0, err0, S0, N0 = _run_reference(tenpy, 10, 16, 3)
        calls = {'lanczos': 0, 'svd': 0}
        plugin.install(tenpy)
        try:
            import tenpy.linalg.krylov_based as kb
            import tenpy.linalg.np_conserved as rnpc
            run, svd = kb.LanczosGroundState.run, rnpc.svd

            def counted_run(self):
                calls['lanczos'] += 1
                return run(self)

            def counted_svd(*a, **k):
                calls['svd'] += 1
                return svd(*a, **k)
            kb.LanczosGroundState.run, rnpc.svd = counted_run, counted_svd
            E1, err1, S1, N1 = _run_reference(tenpy, 10, 16, 3)
        finally:
            plugin.uninstall()
        assert kb.LanczosGroundState.run is not counted_run and rnpc.svd is not counted_svd
    n_updates = 3 * 2 * (10 - 2)
    assert calls['lanczos'] == n_updates and calls['svd'] == n_updates, calls
    np.testing.assert_allclose(E1, E0, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(err1, err0, rtol=0, atol=1e-12)
    assert N1 == N0
    for a, b in zip(S1, S0):
        np.testing.assert_allclose(np.sort(a)[::-1], np.sort(b)[::-1], rtol=0, atol=1e-10)


def test_array_conversion_round_trip(backend):
    """to_device / to_reference: legs (incl. pipes), _qdata, qtotal, labels, blocks survive a round trip exactly."""
    if backend != 'mock':
        pytest.skip("reference tree is not on the GPU box")
    if REF not in sys.path:
        sys.path.insert(0, REF)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        import tenpy
        import tenpy.linalg.np_conserved as rnpc
        from tenpy.linalg import charges as rc
        from tenpy_amd import plugin
        ci = rc.ChargeInfo([1, 3], ['N', 'Z3'])
        rng = np.random.RandomState(7)
        legs = [rc.LegCharge.from_qflat(ci, np.stack([rng.randint(-2, 3, n), rng.randint(0, 3, n)], 1), q).bunch()[1]
                for n, q in ((7, 1), (5, -1), (6, 1))]
        a = rnpc.Array.from_func(rng.standard_normal, legs, qtotal=[1, 2], labels=['a', 'b', 'c'])
        a = a.combine_legs([['a', 'c']], qconj=[-1])
        d = plugin.to_device(a)
        d.test_sanity()
        np.testing.assert_array_equal(d.to_ndarray(), a.to_ndarray())
        back = plugin.to_reference(d, tenpy)
        back.test_sanity()
        assert back.get_leg_labels() == a.get_leg_labels()
        np.testing.assert_array_equal(back._qdata, a._qdata)
        np.testing.assert_array_equal(back.qtotal, a.qtotal)
        for x, y in zip(back._data, a._data):
            np.testing.assert_array_equal(x, y)
        for lb, la in zip(back.legs, a.legs):
            lb.test_equal(la)


def test_plugin_with_mixer_and_qr(backend):
    """Reference DMRG with the density-matrix mixer (``npc.eigh`` of the mixed density matrices through the plugin) and the
    reference's ``MPS.canonical_form`` (``npc.qr`` through the plugin) reproduce the un-patched runs."""
    if backend != 'mock':
        pytest.skip("reference tree is not on the GPU box")
    if REF not in sys.path:
        sys.path.insert(0, REF)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        import tenpy
        from tenpy_amd import plugin
        E0, err0, S0, N0 = _run_reference(tenpy, 8, 12, 3, mixer=True)
        import tenpy.linalg.np_conserved as rnpc
        plugin.install(tenpy)
        try:
            calls = {'eigh': 0, 'qr': 0}
            eigh, qr = rnpc.eigh, rnpc.qr

            def c_eigh(*a, **k):
                calls['eigh'] += 1
                return eigh(*a, **k)

            def c_qr(*a, **k):
                calls['qr'] += 1
                return qr(*a, **k)
            rnpc.eigh, rnpc.qr = c_eigh, c_qr
            E1, err1, S1, N1 = _run_reference(tenpy, 8, 12, 3, mixer=True)
            from tenpy.models.xxz_chain import XXZChain
            from tenpy.networks.mps import MPS
            M = XXZChain({'L': 6, 'Jxx': 1., 'Jz': 1., 'hz': 0., 'bc_MPS': 'finite', 'sort_charge': True})
            psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * 3, bc='finite')
            psi.canonical_form()
        finally:
            plugin.uninstall()
    assert calls['eigh'] > 0 and calls['qr'] > 0
    np.testing.assert_allclose(E1, E0, rtol=1e-9, atol=1e-9)
    for a, b in zip(S1, S0):
        if np.ndim(a) == 1 and np.ndim(b) == 1:
            np.testing.assert_allclose(np.sort(a)[::-1], np.sort(b)[::-1], rtol=0, atol=1e-8)
