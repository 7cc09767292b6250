import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import build_ref
sys.path.insert(0, build_ref.reference_root())
import refsuite_plugin
warnings.simplefilter('ignore')
import numpy as np
from tenpy.models.tf_ising import TFIChain
from tenpy.networks.mps import MPS
from tenpy.algorithms import mpo_evolution
import tenpy.linalg.truncation as tr
import tenpy.linalg.np_conserved as npc
orig = tr._eig_based_svd
def spy(A, **kw):
    U, S, Vd, err, ren = orig(A, **kw)
    print("eig_svd", A.shape, "kept", len(S), "S min", S.min() if len(S) else None, flush=True)
    return U, S, Vd, err, ren
tr._eig_based_svd = spy
M = TFIChain({'J': 1.0, 'g': 1.0, 'L': 16, 'bc_MPS': 'finite', 'conserve': 'parity'})
psi = MPS.from_lat_product_state(M.lat, [['up']])
eng = mpo_evolution.ExpMPOEvolution(psi, M, {'dt': 0.1, 'N_steps': 1, 'order': 1, 'approximation': 'I', 'cbe_min_block_increase': 1,
    'cbe_expand': 0.1, 'use_eig_based_svd': True, 'compression_method': 'variationalQR', 'trunc_params': {'chi_max': 50, 'svd_min': 1e-12}})
for i in range(15):
    eng.run()
    print(i, psi.chi, flush=True)
# direct eigh check
A = np.random.RandomState(0).standard_normal((30, 8)); G = A @ A.T
a = npc.Array.from_ndarray_trivial(G)
w, v = npc.eigh(a)
print("eigh of rank-8 PSD 30x30: ", np.sort(w)[:6], np.sort(np.linalg.eigvalsh(G))[:6])
