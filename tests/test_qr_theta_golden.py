"""decompose_theta_qr_based (QR-based truncation, reference truncation.py:533) vs golden results of the reference,
all flag combinations.  Isometries have a gauge freedom (signs / unitary in degenerate subspaces), so what is
compared is: S (1e-10 of the largest), the reconstructed theta, forms, leg structure, eps, renormalization."""
import numpy as np
import pytest

from helpers import golden, load_array, load_leg, assert_leg_equal
from tenpy_amd.linalg import np_conserved as npc
from tenpy_amd.linalg.truncation import decompose_theta_qr_based


def test_decompose_theta_qr_based(backend):
    recs = golden('qr_theta.pkl')
    assert len(recs) == 8
    for rec in recs:
        theta = load_array(rec['theta'])
        # theta's legs are pipes in the reference: rebuild them so split/labels work
        T_Lc, S, T_Rc, form, err, renorm = decompose_theta_qr_based(
            rec['old_qtotal_L'], rec['old_qtotal_R'], load_leg(rec['old_bond_leg']), theta, rec['move_right'], 0.1, 1,
            rec['eig'], {'chi_max': 20, 'svd_min': 1e-10}, rec['both'], rec['both'])
        assert form == rec['form']
        np.testing.assert_allclose(S, rec['S'], rtol=0, atol=(1e-7 if rec['eig'] else 1e-10) * np.max(rec['S']))
        assert abs(renorm - rec['renorm']) <= 1e-9 * rec['renorm']
        for T, g in ((T_Lc, rec['T_Lc']), (T_Rc, rec['T_Rc'])):
            assert (T is None) == (g is None)
            if T is not None:
                T.test_sanity()
                assert T._labels == g['labels']
                for leg, ld in zip(T.legs, g['legs']):
                    assert_leg_equal(leg, ld, check_pipe=False)
                np.testing.assert_array_equal(T.qtotal, g['qtotal'])
        if rec['both']:
            assert abs(err.eps - rec['eps']) < 1e-10
            if rec['eig']:
                mine = npc.tensordot(T_Lc, T_Rc, ['vR', 'vL']).to_ndarray()
            else:
                mine = npc.tensordot(T_Lc.scale_axis(S, 'vR'), T_Rc, ['vR', 'vL']).to_ndarray()
            gl, gr = rec['T_Lc']['dense'], rec['T_Rc']['dense']
            ref = (gl @ gr) if rec['eig'] else ((gl * rec['S']) @ gr)
            np.testing.assert_allclose(mine, ref, rtol=0, atol=1e-7 if rec['eig'] else 1e-10)
        # the kept isometry really is one
        if rec['move_right']:
            A = T_Lc.to_ndarray()
            np.testing.assert_allclose(A.conj().T @ A, np.eye(A.shape[1]), atol=1e-10)
        else:
            B = T_Rc.to_ndarray()
            np.testing.assert_allclose(B @ B.conj().T, np.eye(B.shape[0]), atol=1e-10)
