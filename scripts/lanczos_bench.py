"""Where does the time of one Lanczos bond solve go at chi=2048?  Synthetic Sz block structure (like matvec_factored_bench.py),
N fixed Lanczos steps through tenpy_amd.linalg.krylov_based.LanczosGroundState; wall time per step, GEMM time from HIP events,
and (under rocprofv3 --kernel-trace --stats) the per-kernel times to compare with."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from tenpy_amd.algorithms.mps_common import TwoSiteH
from tenpy_amd.linalg import np_conserved as npc
from tenpy_amd.linalg.charges import LegCharge
from tenpy_amd.linalg.krylov_based import LanczosGroundState
from tenpy_amd.models.spin_chains import xxz_chain_mpo
from gemm_bench import sectors

chi = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
N = int(sys.argv[3]) if len(sys.argv) > 3 else 8
H = xxz_chain_mpo(8, 1., 1., 0.)
W0, W1 = H.get_W(3), H.get_W(4)
ch = W0.chinfo
q, n = sectors(chi)
bond = LegCharge.from_qind(ch, np.concatenate([[0], np.cumsum(n)]), q.reshape(-1, 1), qconj=+1)
rng = np.random.default_rng(0)
rnd = lambda sh: rng.standard_normal(sh)
LP = npc.Array.from_func(rnd, [bond, W0.get_leg('wL').conj(), bond.conj()], labels=['vR*', 'wR', 'vR'])
RP = npc.Array.from_func(rnd, [bond, W1.get_leg('wR').conj(), bond.conj()], labels=['vL', 'wL', 'vL*'])
# hermitian operator: symmetrise the environments in their bond indices
LP = LP + LP.conj().itranspose(['vR*', 'wR*', 'vR']).ireplace_label('wR*', 'wR') if False else LP
eff = TwoSiteH(None, 3, tensors=(LP, RP, W0, W1))
p = W0.get_leg('p')
theta = npc.Array.from_func(rnd, [bond, p, p, bond.conj()], labels=['vL', 'p0', 'p1', 'vR'])
theta.iscale_prefactor(1. / npc.norm(theta))
opts = {'N_min': N, 'N_max': N}
import warnings, logging
logging.disable(logging.WARNING)
for _ in range(2):
    LanczosGroundState(eff, theta, opts).run()
torch.cuda.synchronize()
npc.gemm_timer.reset()
npc.gemm_timer.enabled = True
t0 = time.time()
for _ in range(reps):
    E, th, n_it = LanczosGroundState(eff, theta, opts).run()
torch.cuda.synchronize()
dt = (time.time() - t0) / reps
ms = npc.gemm_timer.collect()
print("lanczos chi=%d N=%d: %.3f ms per run, %.3f ms per step; GEMM %.3f ms per step (%.1f TFLOP/s); pipelined=%s" % (
    chi, n_it, dt * 1e3, dt * 1e3 / n_it, ms / reps / n_it, npc.gemm_timer.flops / (ms * 1e-3) / 1e12,
    os.environ.get('TPA_LANCZOS_PIPELINED', '1')), flush=True)
