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
            # the middle MPO indices (S+, S-, Sz channels) may be ordered differently: compare as sets of matrices
            if not np.allclose(mine, ref, atol=1e-13):
                a = sorted([np.round(mine[x, y], 12).tobytes() for x in range(mine.shape[0]) for y in range(mine.shape[1])])
                b = sorted([np.round(ref[x, y], 12).tobytes() for x in range(ref.shape[0]) for y in range(ref.shape[1])])
                assert a == b
        psi = MPS.from_product_state([p] * L, [1, 0] * (L // 2))
        eng = ExpMPOEvolution(psi, H, dict(rec['options']))
        for step in rec['steps']:
            eng.run()
            assert list(psi.chi) == step['chi']
            np.testing.assert_allclose(psi.entanglement_entropy(), step['S'], rtol=0, atol=1e-10)
            np.testing.assert_allclose(psi.expectation_value(np.diag([-0.5, 0.5])), step['Sz'], rtol=0, atol=1e-10)
            assert abs(psi.norm - step['norm']) < 1e-10
        assert abs(eng.trunc_err.eps - rec['trunc_err']) < 1e-11
