"""Real-time TEBD (order 2) after a global quench vs the reference's TEBDEngine (golden): complex128 path
(tensordot with the bond gate, scale_axis, combine_legs, block SVD, split_legs, tensordot with V.conj())."""
import numpy as np
import pytest

from helpers import golden
from tenpy_amd.algorithms.tebd import TEBDEngine
from tenpy_amd.linalg import np_conserved as npc
from tenpy_amd.models.spin_chains import spin_half_leg
from tenpy_amd.networks.mps import MPS


@pytest.mark.parametrize("batch", [True, False])
@pytest.mark.parametrize("name", ['tfi_quench_L10_parity', 'tfi_quench_L10_None'])
def test_tebd_quench(backend, name, batch):
    """``batch``: the independent bonds of a half-step decomposed in one batched block SVD (``np_conserved.svd_batched``, the default)
    or bond by bond -- both against the reference's run."""
    rec = [r for r in golden('tebd.pkl') if r['name'] == name][0]
    L = rec['L']
    _, p = spin_half_leg(rec['conserve'])
    up = dict(rec['state_labels'])['up']
    psi = MPS.from_product_state([p] * L, [up] * L, dtype=np.complex128)
    eng = TEBDEngine(psi, rec['h_bond'], {'dt': rec['dt'], 'trunc_params': {'chi_max': rec['chi'], 'svd_min': 1.e-10}, 'batch_bonds': batch})
    sz = np.diag([1., -1.]) if up == 0 else np.diag([-1., 1.])
    for step in range(len(rec['chi_t'])):
        eng.evolve_step_order2()
        assert max(psi.chi) == rec['chi_t'][step]
        np.testing.assert_allclose(psi.entanglement_entropy(), rec['S_t'][step], rtol=0, atol=1e-10)
    np.testing.assert_allclose(np.sort(psi.get_SL(L // 2))[::-1], np.sort(rec['S_mid'])[::-1], rtol=0, atol=1e-10)
    # <sigma^z_i> from the B-form tensors
    ev = []
    for i in range(L):
        th = psi.get_B(i, 'B').scale_axis(psi.get_SL(i), 'vL')
        d = th.to_ndarray()
        ev.append(np.real(np.einsum('apb,pq,aqb->', d.conj(), sz, d)))
    np.testing.assert_allclose(ev, rec['sigmaz_t'][-1], rtol=0, atol=1e-10)
    assert psi.get_B(0, None).dtype == np.complex128


@pytest.mark.parametrize("batch", [True, False])
@pytest.mark.parametrize("name", ['tfi_quench_L10_parity', 'tfi_quench_L10_None'])
def test_qr_tebd_quench(backend, name, batch):
    """QR-based TEBD (two tensordots + two block QRs + SVD of the small bond matrix) vs the reference's
    QRBasedTEBDEngine on the same quench.  ``batch``: the bond matrices of a half-step in one batched block SVD (the default)."""
    from tenpy_amd.algorithms.tebd import QRBasedTEBDEngine
    rec = [r for r in golden('tebd.pkl') if r['name'] == name][0]
    L = rec['L']
    _, p = spin_half_leg(rec['conserve'])
    up = dict(rec['state_labels'])['up']
    psi = MPS.from_product_state([p] * L, [up] * L, dtype=np.complex128)
    eng = QRBasedTEBDEngine(psi, rec['h_bond'], {'dt': rec['dt'], 'cbe_expand': 0.5, 'batch_bonds': batch,
                                                 'trunc_params': {'chi_max': rec['chi'], 'svd_min': 1.e-10}})
    for step in range(len(rec['chi_qr'])):
        eng.evolve_step_order2()
        assert max(psi.chi) == rec['chi_qr'][step]
        np.testing.assert_allclose(psi.entanglement_entropy(), rec['S_qr'][step], rtol=0, atol=1e-9)


def test_qr_tebd_eig_route_batched_equals_bond_by_bond(backend):
    """``use_eig_based_svd``: the Hermitian eigenproblems of the bond matrices of a half-step in ONE ``eigh_batched`` call (the default)
    give the state of the bond-by-bond loop -- same bond dimensions, Schmidt values and truncation error (reference
    ``QRBasedTEBDEngine.update_bond`` with ``use_eig_based_svd``, tebd.py:685-738 / truncation.py:473-530)."""
    from tenpy_amd.algorithms.tebd import QRBasedTEBDEngine
    rec = [r for r in golden('tebd.pkl') if r['name'] == 'tfi_quench_L10_parity'][0]
    L = rec['L']
    _, p = spin_half_leg(rec['conserve'])
    up = dict(rec['state_labels'])['up']
    out = []
    for batch in (True, False):
        psi = MPS.from_product_state([p] * L, [up] * L, dtype=np.complex128)
        eng = QRBasedTEBDEngine(psi, rec['h_bond'], {'dt': rec['dt'], 'cbe_expand': 0.5, 'batch_bonds': batch, 'use_eig_based_svd': True,
                                                     'trunc_params': {'chi_max': rec['chi'], 'svd_min': 1.e-10}})
        for step in range(len(rec['chi_qr'])):
            eng.evolve_step_order2()
        out.append((list(psi.chi), [np.asarray(psi.get_SL(i)) for i in range(1, L)], eng.trunc_err.eps, psi.entanglement_entropy()))
    assert out[0][0] == out[1][0]
    for a, b in zip(out[0][1], out[1][1]):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-12)
    assert abs(out[0][2] - out[1][2]) <= 1e-14
    # and against the reference's QR engine (SVD of the bond matrix): the eigenvalue route loses the tiny values, not the entropy
    np.testing.assert_allclose(out[0][3], rec['S_qr'][len(rec['chi_qr']) - 1], rtol=0, atol=1e-7)


def test_batch_group_size_respects_the_memory_cap(backend):
    """``batch_bonds=True`` batches a whole half-step only as far as the SVD work areas (~24x the dense theta per bond) fit the cap."""
    from tenpy_amd.algorithms.tebd import batch_group_size
    _, p = spin_half_leg('parity')
    psi = MPS.from_product_state([p] * 10, [0] * 10, dtype=np.complex128)
    bonds = [1, 3, 5, 7, 9]
    per_bond = 24 * 16 * (2 * 1) * (2 * 1)               # chi = 1 everywhere, d = 2
    assert batch_group_size(psi, bonds, cap=10**9) == 5
    assert batch_group_size(psi, bonds, cap=3 * per_bond) == 3
    assert batch_group_size(psi, bonds, cap=1) == 1
    assert batch_group_size(psi, [], cap=1) == 1
