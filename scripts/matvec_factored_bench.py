"""The Lanczos matvec in its factored form (LP . theta -> W0 W1 blockwise -> . RP) on a synthetic chi-sized Sz block
structure: timing and -- under rocprofv3 --pmc -- the L2-miss traffic of its two GEMM launches (profiles/)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tenpy_amd.algorithms.mps_common import TwoSiteH
from tenpy_amd.linalg import np_conserved as npc
from tenpy_amd.linalg.charges import LegCharge
from tenpy_amd.models.spin_chains import xxz_chain_mpo
from gemm_bench import sectors

chi = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
H = xxz_chain_mpo(8, 1., 1., 0.)
W0, W1 = H.get_W(3), H.get_W(4)
ch = W0.chinfo
q, n = sectors(chi)
bond = LegCharge.from_qind(ch, np.concatenate([[0], np.cumsum(n)]), q.reshape(-1, 1), qconj=+1)   # (bra) bond leg
rnd = lambda sh: np.random.standard_normal(sh)
wR_LP = W0.get_leg('wL').conj()
wL_RP = W1.get_leg('wR').conj()
LP = npc.Array.from_func(rnd, [bond, wR_LP, bond.conj()], labels=['vR*', 'wR', 'vR'])
RP = npc.Array.from_func(rnd, [bond, wL_RP, bond.conj()], labels=['vL', 'wL', 'vL*'])
eff = TwoSiteH(None, 3, tensors=(LP, RP, W0, W1))
assert eff.factored
p = W0.get_leg('p')
theta = npc.Array.from_func(rnd, [bond, p, p, bond.conj()], labels=['vL', 'p0', 'p1', 'vR'])
knobs = sys.argv[3:] or ['0']      # TPA_GEMM_SPLIT_K values to compare ("max parts,target tiles,min K per part")
ref = None
for knob in knobs:
    npc.GEMM_SPLIT_K = tuple(int(x) for x in knob.split(','))
    npc._plan_cache.clear()
    eff._fplans = None
    for _ in range(2):
        out = eff.matvec(theta)
    torch.cuda.synchronize()
    npc.gemm_timer.reset()
    npc.gemm_timer.enabled = True
    t0 = time.time()
    for _ in range(reps):
        out = eff.matvec(theta)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / reps
    ms = npc.gemm_timer.collect()
    gt = npc.gemm_timer
    if ref is None:
        ref = out
    fp = eff._fplans
    desc = ["%d tiles%s" % (pl.n_tiles, "" if pl.sk is None else " -> %d in %d parts" % (pl.sk.n_tiles, len(pl.sk.tasks_host)))
            for pl in (fp['p1'], fp['p2'])]
    print("factored matvec chi=%d split-K %s: %.3f ms per matvec, GEMM %.3f ms (%.1f TFLOP/s, %d launches), flops %.3e, min bytes %.3e; "
          "step 1 %s, step 2 %s; |out - out(first knob)| / |out| = %.1e" % (
              chi, knob, dt * 1e3, ms / reps, gt.flops / (ms * 1e-3) / 1e12, gt.n_launch // reps, eff.flops_per_matvec,
              eff.bytes_per_matvec, desc[0], desc[1], npc.norm(out - ref) / npc.norm(ref)), flush=True)
