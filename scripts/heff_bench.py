"""Time the pieces of TwoSiteH.combine_Heff (LHeff = LP.W0 + leg fusion) on the chi-sized Sz sector structure."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tenpy_amd.linalg import np_conserved as npc
from tenpy_amd.linalg.charges import ChargeInfo, LegCharge
from tenpy_amd.models.spin_chains import xxz_chain_mpo
from gemm_bench import sectors

chi = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
H = xxz_chain_mpo(8, 1., 1., 0.)
W0 = H.get_W(3).replace_labels(['p', 'p*'], ['p0', 'p0*'])
ch = W0.chinfo
q, n = sectors(chi)
vR = LegCharge.from_qind(ch, np.concatenate([[0], np.cumsum(n)]), q.reshape(-1, 1), qconj=-1)
wR = W0.get_leg('wL').conj()
rnd = lambda sh: np.random.standard_normal(sh)
LP = npc.Array.from_func(rnd, [vR.conj(), wR, vR], labels=['vR*', 'wR', 'vR'])


def timeit(f, reps=5):
    f()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        r = f()
    torch.cuda.synchronize()
    return (time.time() - t0) / reps * 1e3, r


t_td, LHeff = timeit(lambda: npc.tensordot(LP, W0, axes=['wR', 'wL']))
pipeL = LHeff.make_pipe(['vR*', 'p0'], qconj=+1)
t_cl, LH2 = timeit(lambda: LHeff.combine_legs([['vR*', 'p0'], ['vR', 'p0*']], pipes=[pipeL, pipeL.conj()], new_axes=[0, 2]))
nbytes = LH2._arena.numel() * 8
print("chi=%d: tensordot(LP, W0) %.3f ms, combine_legs %.3f ms; LHeff %.1f MB (ideal write+read at 4 TB/s: %.3f ms)" % (
    chi, t_td, t_cl, nbytes / 1e6, 2 * nbytes / 4e12 * 1e3), flush=True)
# plan-cached replays only (no host planning)
plan, a_use, b_use = npc.plan_tensordot(LP, W0, axes=['wR', 'wL'])
t_apply, _ = timeit(lambda: plan.apply(a_use, b_use))
print("   plan.apply alone %.3f ms (%d gemms, %d tiles); transposed operand copy needed: %s" % (
    t_apply, plan.n_gemm, plan.n_tiles, a_use is not LP), flush=True)
from tenpy_amd.algorithms import mps_common
t_fused, res = timeit(lambda: mps_common._fused_heff(LP, W0, True))
print("   fused lincomb builder %.3f ms (%s)" % (t_fused, 'ok' if res is not None else 'not applicable'), flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    mps_common._fused_heff(LP, W0, True)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(12)
