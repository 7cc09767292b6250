"""The reference's iDMRG benchmark on the device: ``tests/benchmark/dmrg_infinite.py`` (spin-2 chain, D = 0.3, Sz conserved,
unit cell L = ``legs`` = 10, chi = ``size``, Lanczos N_min = N_max = 10; warm-up = optimisation sweeps alternating with
environment sweeps, timed = 10 optimisation sweeps).

    python scripts/bench_idmrg.py [chi] [L] [warmup_pairs] [timed_sweeps]

Prints seconds per sweep (2 L bond updates) and the energy per site.  Needs a GPU; no CPU fallback.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tenpy_amd.algorithms.dmrg import TwoSiteDMRGEngine
from tenpy_amd.models.spin_chains import spin_S_leg, spin_chain_mpo
from tenpy_amd.networks.mps import MPS

chi = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 30        # the reference uses 100 pairs; chi saturates much earlier
timed = int(sys.argv[4]) if len(sys.argv) > 4 else 10

H = spin_chain_mpo(L, S=2., D=0.3, bc='infinite')
_, p = spin_S_leg(2.)
psi = MPS.from_product_state([p] * L, ([p.ind_len - 1, 0] * L)[:L], bc='infinite')
eng = TwoSiteDMRGEngine(psi, H, {'trunc_params': {'chi_max': chi, 'svd_min': 1.e-45}, 'lanczos_params': {'N_min': 10, 'N_max': 10}})
t0 = time.time()
for i in range(warm):
    eng.sweep()
    eng.sweep(optimize=False)
torch.cuda.synchronize()
print("warm-up: %d sweep pairs in %.1f s, chi = %s" % (warm, time.time() - t0, max(psi.chi)), flush=True)
n0 = len(eng.update_stats['E_total'])
torch.cuda.synchronize()
t0 = time.time()
for i in range(timed):
    eng.sweep()
torch.cuda.synchronize()
dt = (time.time() - t0) / timed
Es, ages = eng.update_stats['E_total'], eng.update_stats['age']
k = min(1 + 2 * L, len(ages) - n0)
print("chi=%d L=%d: %.4f s per sweep (%d bond updates, %.2f ms each), E/site = %.10f, N_lanczos = %.1f"
      % (chi, L, dt, 2 * L, 1e3 * dt / (2 * L), (Es[-1] - Es[-k]) / (ages[-1] - ages[-k]), np.mean(eng.update_stats['N_lanczos'][n0:])))
