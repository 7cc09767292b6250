"""TeNPy ITSELF on BASELINE config 5 (the reference of ``bench.py --config tebd1024``): the reference's ``TEBDEngine``
(tenpy/algorithms/tebd.py:416, order 2, dt = 0.05, chi_max = chi, svd_min = 1e-12) on the SAME synthetic state as the bench
(scripts/tebd_state.py: seeded random right-canonical MPS, TFI chain L = 64 with parity conservation, complex128), with the
compiled ``_npc_helper`` on the host cores of the build container.  Records, after every step, what the bench line reports
(accumulated truncation error, norm factor, entanglement entropy of the centre bond, largest Schmidt values) and the
time of the step: ``profiles/r03_cpu_reference_tebd.json``, which ``bench.py`` quotes as ``cpu_baseline`` (offline) and compares
its own values with (``tebd_parity``).

    python scripts/cpu_reference_tebd.py [chi] [L] [n_steps]          (defaults 1024 64 3)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from oracle import build_ref  # noqa: E402

# TPA_TEBD_ON_DEVICE=1: the SAME script with TeNPy's engines on the MI355X through the module form (install(fused=True)): what
# `profiles/r04_module_form_tebd*.json` hold
DEVICE = bool(os.environ.get('TPA_TEBD_ON_DEVICE'))
if DEVICE:
    sys.path.insert(0, build_ref.reference_root())
    import tenpy_amd.install as ti
    ti.install(fused=os.environ.get('TPA_TEBD_FUSED', '1') != '0')
else:
    build_ref.load()
import numpy as np  # noqa: E402
import tenpy  # noqa: E402
import tenpy.linalg.np_conserved as npc  # noqa: E402
from tenpy.algorithms import tebd  # noqa: E402
from tenpy.linalg.charges import LegCharge, LegPipe  # noqa: E402
from tenpy.models.tf_ising import TFIChain  # noqa: E402
from tenpy.networks.mps import MPS  # noqa: E402
from tenpy.tools import optimization  # noqa: E402
import tebd_state  # noqa: E402

assert DEVICE or optimization.have_cython_functions, "compiled _npc_helper not active"
optimization.set_level(3)
chi = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
OUT = os.environ.get('TPA_CPU_REF_OUT') or os.path.join(ROOT, 'profiles', 'r03_cpu_reference_tebd.json')

M = TFIChain({'L': L, 'J': 1., 'g': 1.5, 'bc_MPS': 'finite', 'conserve': 'parity', 'sort_charge': True})
sites = M.lat.mps_sites()
p = sites[0].leg
t0 = time.time()
Bs, Ss = tebd_state.random_right_canonical_tensors(npc, LegCharge, LegPipe, p, L, chi, np.complex128, seed=1)
psi = MPS(sites, Bs, Ss, bc='finite', form='B')
t_state = time.time() - t0
print("state built in %.1f s, chi = %s" % (t_state, max(psi.chi)), flush=True)
QR = os.environ.get('TPA_CPU_REF_ENGINE', 'svd') == 'qr'       # the like-for-like reference of `bench.py --config tebd1024 --qr`
if DEVICE:
    OUT = os.environ.get('TPA_CPU_REF_OUT') or os.path.join(ROOT, 'gpurun_out', 'r04_module_form_tebd%s.json' % ('_qr' if QR else ''))
EIG = bool(os.environ.get('TPA_CPU_REF_EIG_SVD'))          # round 6: the like-for-like reference of `bench.py --config tebd1024 --qr --eig-svd`
if QR:
    if not DEVICE:
        OUT = os.environ.get('TPA_CPU_REF_OUT') or os.path.join(ROOT, 'profiles', 'r06_cpu_reference_tebd_qr_eig.json' if EIG else 'r04_cpu_reference_tebd_qr.json')
    eng = tebd.QRBasedTEBDEngine(psi, M, {'order': 2, 'dt': 0.05, 'N_steps': 1, 'compute_err': True, 'cbe_expand': 0.1,
                                          'use_eig_based_svd': EIG, 'trunc_params': {'chi_max': chi, 'svd_min': 1e-12}})
else:
    eng = tebd.TEBDEngine(psi, M, {'order': 2, 'dt': 0.05, 'N_steps': 1, 'trunc_params': {'chi_max': chi, 'svd_min': 1e-12}})
steps = []
def _sync():
    if DEVICE:
        import torch
        torch.cuda.synchronize()


for k in range(n_steps):
    _sync()
    t0 = time.time()
    eng.run()
    _sync()
    dt = time.time() - t0
    S = np.asarray(psi.get_SL(L // 2))
    steps.append({"step": k + 1, "s": dt, "trunc_err_eps": float(sum(e.eps for e in eng._trunc_err_bonds)),
                  "engine_trunc_err_eps": float(eng.trunc_err.eps),     # run() counts every step twice (algorithm.py:440 on top of tebd.py:370)

                  "S_mid_entropy": float(-np.sum(S ** 2 * np.log(S ** 2 + 1e-300))), "chi_mid": int(len(S)),
                  "schmidt_top8": [float(x) for x in np.sort(S)[::-1][:8]]})
    print(steps[-1], flush=True)
    import scipy
    out = {"engine": ("QRBasedTEBDEngine (tebd.py:622; cbe_expand 0.1, compute_err True, use_eig_based_svd %s)" % EIG) if QR else "TEBDEngine",
           "what": "TeNPy %s TEBD engine (order 2, dt 0.05, one step per run(), svd_min 1e-12) on the synthetic state of bench.py --config tebd1024: "
                   "TFIChain L=%d J=1 g=1.5 conserve=parity, random right-canonical MPS chi=%d complex128 seed 1 (scripts/tebd_state.py)"
                   % (tenpy.__version__, L, chi),
           "where": ("MI355X, module form: tenpy_amd.install.install(fused=%s), TeNPy's engine unmodified" % (os.environ.get('TPA_TEBD_FUSED', '1') != '0')) if DEVICE else "host cores",
           "cores": os.cpu_count(), "blas_threads": os.environ.get('OMP_NUM_THREADS', 'default (all cores)'),
           "helper": "compiled _npc_helper (oracle/_ref), scipy.linalg.cython_blas", "numpy": np.__version__, "scipy": scipy.__version__,
           "bond_updates_per_step": 3 * (L // 2) - 1 if L % 2 == 0 else None, "L": L, "chi": chi, "steps": steps,
           "s_per_step_best": min(s["s"] for s in steps)}
    with open(OUT, 'w') as f:
        json.dump(out, f, indent=1)
