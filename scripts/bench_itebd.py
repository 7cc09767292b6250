"""The reference's TEBD benchmark on the device: ``tests/benchmark/tebd_infinite.py`` (infinite spin-2 chain, D = 0.3,
Sz conserved, unit cell L = 10, chi = ``size``, order 2, N_steps = 5, dt = 0.1; warm-up runs until every bond has the full
bond dimension, timed = one ``run()`` = 5 time steps).

    python scripts/bench_itebd.py [chi] [L]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tenpy_amd.algorithms.tebd import TEBDEngine
from tenpy_amd.models.spin_chains import spin_S_leg, spin_chain_h_bonds
from tenpy_amd.networks.mps import MPS

chi = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10
_, p = spin_S_leg(2.)
d = p.ind_len
psi = MPS.from_product_state([p] * L, ([d - 1, 0] * L)[:L], dtype=np.complex128, bc='infinite')
eng = TEBDEngine(psi, spin_chain_h_bonds(L, S=2., D=0.3, bc='infinite'),
                 {'order': 2, 'N_steps': 5, 'dt': 0.1, 'trunc_params': {'chi_max': chi, 'svd_min': 1.e-45}})
t0 = time.time()
for i in range(5 + int(np.log(chi) / np.log(d))):
    eng.run_evolution()
torch.cuda.synchronize()
print("warm-up %.1f s, chi = %s" % (time.time() - t0, psi.chi), flush=True)
best = None
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.time()
    eng.run_evolution()
    torch.cuda.synchronize()
    dt = time.time() - t0
    best = dt if best is None else min(best, dt)
print("chi=%d L=%d: %.4f s per run() = 5 steps of order 2 (%d bond updates, %.2f ms each), min chi %d"
      % (chi, L, best, 11 * L // 2 * 1, 1e3 * best / (5.5 * L), min(psi.chi)))
