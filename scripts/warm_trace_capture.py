"""Dev harness (CPU, numpy emulation of the device): run the stand-alone two-site DMRG driver to a converged state and capture the
matrices the WARM-started block Jacobi is handed (rows of W = Bq X^H per charge block), for the offline emulation of the iteration
in scripts/warm_trace_emulate.py.  Usage: python scripts/warm_trace_capture.py L chi n_sweeps out.npz"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np


class _MP:
    def setattr(self, obj, name, value):
        setattr(obj, name, value)


import mock_device  # noqa: E402
mock = mock_device.install(_MP())
from tenpy_amd.algorithms.dmrg import TwoSiteDMRGEngine  # noqa: E402
from tenpy_amd.linalg import _svd_warm  # noqa: E402
from tenpy_amd.linalg import np_conserved as npc  # noqa: E402
from tenpy_amd.models.spin_chains import spin_half_leg, xxz_chain_mpo  # noqa: E402
from tenpy_amd.networks.mps import MPS  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 40
chi = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 8
out = sys.argv[4] if len(sys.argv) > 4 else '/tmp/warm_trace.npz'

captured = {}
state = {'alg': 0, 'on': False, 'n': 0}
orig_batch = mock.tpa_svd_batch
orig_set = mock.tpa_svd_set_algorithm


def set_alg(a):
    state['alg'] = a
    return orig_set(a)


def batch(code, jobs_p, n_jobs, a_p, u_p, s_p, vh_p, work_p, wb, max_sweeps, tol, sweeps_p, stream):
    if state['on'] and (state['alg'] & 512) and code == 0:
        jobs = mock_device._host(jobs_p, (n_jobs, 8))
        A = mock_device.REG.view(a_p, np.float64)
        big = int(np.argmax(jobs[:, 1]))
        a_off, m, n = (int(x) for x in jobs[big, :3])
        if m >= 48:
            captured['W%03d_%s' % (state['n'], _svd_warm.last_kind)] = A[a_off:a_off + m * n].reshape(m, n).copy()
            state['n'] += 1
    return orig_batch(code, jobs_p, n_jobs, a_p, u_p, s_p, vh_p, work_p, wb, max_sweeps, tol, sweeps_p, stream)


mock.tpa_svd_batch = batch
mock.tpa_svd_set_algorithm = set_alg

H = xxz_chain_mpo(L, 1., 1., 0.)
_, p = spin_half_leg('Sz')
psi = MPS.from_product_state([p] * L, [1, 0] * (L // 2))
chi_list = {0: 32, 2: 64, 3: chi}
eng = TwoSiteDMRGEngine(psi, H, {'chi_list': chi_list, 'trunc_params': {'chi_max': chi, 'svd_min': 1e-14},
                                 'lanczos_params': {'N_min': 8, 'N_max': 8}})
import time
for s in range(ns):
    state['on'] = s >= ns - 2
    t = time.time()
    eng.sweep()
    print("sweep %d E=%.14f chi=%d t=%.1fs" % (s, eng.sweep_stats['E'][-1], eng.sweep_stats['max_chi'][-1], time.time() - t),
          {k: _svd_warm.stats[k] for k in ('warm_calls', 'sketch_calls', 'cold_calls')}, flush=True)
# keep the largest ones
keys = sorted(captured, key=lambda k: -captured[k].shape[0])[:24]
np.savez(out, **{k: captured[k] for k in keys})
print('saved', len(keys), 'matrices, shapes', sorted({captured[k].shape for k in keys}))
