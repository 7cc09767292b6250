"""Real-time TEBD after a global quench (BASELINE config 5): TFI chain, parity conserved, order 2.
python scripts/run_tebd.py L chi n_steps [qr]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tenpy_amd.algorithms.tebd import TEBDEngine, QRBasedTEBDEngine
from tenpy_amd.linalg import np_conserved as npc
from tenpy_amd.models.spin_chains import spin_half_leg
from tenpy_amd.networks.mps import MPS

L = int(sys.argv[1]) if len(sys.argv) > 1 else 64
chi = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
use_qr = len(sys.argv) > 4 and sys.argv[4] == 'qr'
J, g = 1., 1.5
_, p = spin_half_leg('parity')
sx = np.array([[0., 1.], [1., 0.]])
sz = np.diag([-1., 1.])
I2 = np.eye(2)
h_bonds = [None]
for i in range(1, L):
    gl = g if i - 1 == 0 else g / 2
    gr = g if i == L - 1 else g / 2
    h = -J * np.kron(sx, sx) - gl * np.kron(sz, I2) - gr * np.kron(I2, sz)
    h_bonds.append(h.reshape(2, 2, 2, 2))
psi = MPS.from_product_state([p] * L, [1] * L, dtype=np.complex128)
cls = QRBasedTEBDEngine if use_qr else TEBDEngine
eng = cls(psi, h_bonds, {'dt': 0.05, 'cbe_expand': 0.1, 'compute_err': False, 'trunc_params': {'chi_max': chi, 'svd_min': 1e-12}})
for step in range(n_steps):
    torch.cuda.synchronize()
    t0 = time.time()
    eng.evolve_step_order2()
    torch.cuda.synchronize()
    if step % max(1, n_steps // 10) == 0 or step == n_steps - 1:
        print("step %d t=%.2f chi=%d S=%.4f  %.3f s/step  svd %s" % (step, eng.evolved_time, max(psi.chi), max(psi.entanglement_entropy()),
                                                                    time.time() - t0, npc.svd_stats), flush=True)
        npc.svd_stats.update(calls=0, sweeps=0, max_block=0)
