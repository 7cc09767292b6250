"""``ExpMPOEvolution`` (W_II time evolution operators applied as MPOs + SVD compression) vs the reference
(tests/golden/make_golden.py:gen_mpo_evolution): the W_II tensors themselves, then entropies / bond dimensions / <Sz> / norm
after every run."""
import numpy as np

from helpers import golden
from tenpy_amd.algorithms.mpo_evolution import ExpMPOEvolution
from tenpy_amd.models.spin_chains import spin_half_leg, xxz_chain_mpo
from tenpy_amd.networks.mps import MPS


def test_exp_mpo_evolution(backend):
    for rec in golden('mpo_evolution.pkl'):
        L = rec['L']
        H = xxz_chain_mpo(L, rec['Jxx'], rec['Jz'], rec['hz'])
        _, p = spin_half_leg('Sz')
        U = H.make_U_II(-0.05j)
        for i in range(L):
            mine = U.get_W(i).transpose(['wL', 'wR', 'p', 'p*']).to_ndarray()
            ref = rec['W_II'][i]
            assert mine.shape == ref.shape
            # (the S+ / S- / Sz channels of the MPO bond may be numbered differently: compare the multiset of operator entries)
            def canon(W):
                v = W.reshape(-1, W.shape[2] * W.shape[3])
                key = np.round(np.concatenate([v.real, v.imag], axis=1), 9) + 0.
                return v[np.lexsort(key.T[::-1])]
            np.testing.assert_allclose(canon(mine), canon(ref), rtol=0, atol=1e-12)
        psi = MPS.from_product_state([p] * L, [1, 0] * (L // 2))
        eng = ExpMPOEvolution(psi, H, dict(rec['options']))
        for step in rec['steps']:
            eng.run()
            assert list(psi.chi) == step['chi']
            np.testing.assert_allclose(psi.entanglement_entropy(), step['S'], rtol=0, atol=1e-10)
            np.testing.assert_allclose(psi.expectation_value(np.diag([-0.5, 0.5])), step['Sz'], rtol=0, atol=1e-10)
            assert abs(psi.norm - step['norm']) < 1e-10
        assert abs(eng.trunc_err.eps - rec['trunc_err']) < 1e-11
