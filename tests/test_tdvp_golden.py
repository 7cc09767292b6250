"""TDVP (two-site, then single-site on the same state) vs the reference's engines after a quench from the Neel state
(tests/golden/make_golden.py:gen_tdvp): entropies, bond dimensions, <Sz_i>, norm, evolved time and accumulated truncation
error after every ``run()``; total energy conserved."""
import numpy as np

from helpers import golden
from tenpy_amd.algorithms.tdvp import SingleSiteTDVPEngine, TwoSiteTDVPEngine
from tenpy_amd.linalg import np_conserved as npc
from tenpy_amd.models.spin_chains import spin_half_leg, xxz_chain_mpo
from tenpy_amd.networks.mps import MPS


def _sz(psi):
    ev = []
    for i in range(psi.L):
        th = psi.get_B(i, 'Th').to_ndarray()          # (vL, p, vR), p: index 0 = down, 1 = up
        ev.append(np.real(np.einsum('apb,p,apb->', th.conj(), np.array([-0.5, 0.5]), th)))
    return np.array(ev)


def test_tdvp_quench(backend):
    rec = golden('tdvp.pkl')[0]
    L = rec['L']
    H = xxz_chain_mpo(L, rec['Jxx'], rec['Jz'], rec['hz'])
    _, p = spin_half_leg('Sz')
    psi = MPS.from_product_state([p] * L, [1, 0] * (L // 2))
    eng2 = TwoSiteTDVPEngine(psi, H, dict(rec['options']))
    eng1 = None
    for step in rec['steps']:
        if step['engine'] == 'two':
            eng = eng2
        else:
            if eng1 is None:
                eng1 = SingleSiteTDVPEngine(psi, H, dict(rec['options']))
            eng = eng1
        eng.run()
        assert list(psi.chi) == step['chi']
        np.testing.assert_allclose(psi.entanglement_entropy(), step['S'], rtol=0, atol=1e-10)
        np.testing.assert_allclose(_sz(psi), step['Sz'], rtol=0, atol=1e-10)
        np.testing.assert_allclose(psi.expectation_value(np.diag([-0.5, 0.5])), step['Sz'], rtol=0, atol=1e-10)
        assert abs(psi.norm - step['norm']) < 1e-10
        assert abs(eng.evolved_time - step['t']) < 1e-14
        assert abs(eng.trunc_err.eps - step['trunc_err']) < 1e-11
    assert psi.get_B(0, None).dtype == np.complex128
    # energy of the evolved state (conserved by TDVP up to truncation) from a fresh environment
    from tenpy_amd.networks.mpo import MPOEnvironment
    for i in range(L):                                  # bring the state to B form for the environment contraction
        psi.set_B(i, psi.get_B(i, 'B'), form='B')
    E = np.real(MPOEnvironment(psi, H).full_contraction(L // 2 - 1))
    assert abs(E - rec['E']) < 1e-9
