"""UNMODIFIED TeNPy on the MI355X: ``tenpy.algorithms.dmrg.TwoSiteDMRGEngine`` (the reference's own engine, mixers, MPS / MPO /
environment classes) on the device mirror, ``tenpy_amd.install.install(fused=True)``, with bench.py's protocol
(VERDICT r2 task 2): Neel state, chi ramp 64, 64, 128, ..., chi/2 WITH the density-matrix mixer on (``--mixer``; covers
``DensityMatrixMixer`` ``mps_common.py:1903-2079`` and ``eigh`` on hardware), then sweeps at the target chi with Lanczos N = 8.

    python scripts/module_form_bench.py [--chi 2048] [--L 100] [--sweeps 3] [--combine 0|1] [--mixer 0|1] [--tebd]

Prints one JSON line: seconds of every sweep, energies (compare: profiles/r02_cpu_reference.json, TeNPy on the CPU), which
TwoSiteH class ran how often, SVD statistics.  The reference comes from /root/reference or oracle/_ref/tenpy_ref.zip."""
import argparse
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--chi', type=int, default=2048)
ap.add_argument('--L', type=int, default=100)
ap.add_argument('--sweeps', type=int, default=3)
ap.add_argument('--combine', type=int, default=0)
ap.add_argument('--mixer', type=int, default=1)
ap.add_argument('--fused', type=int, default=1)
ap.add_argument('--tebd', action='store_true')
args = ap.parse_args()

from oracle import build_ref  # noqa: E402
root = build_ref.reference_root()
assert root is not None, "no reference tree / archive"
sys.path.insert(0, root)
import tenpy_amd.install as ti  # noqa: E402
ti.install()
warnings.simplefilter('ignore')
import torch  # noqa: E402
import tenpy  # noqa: E402
if args.fused:
    ti.use_fused_callers()
import tenpy.linalg.np_conserved as npc  # noqa: E402
import tenpy_amd.linalg.np_conserved as mirror  # noqa: E402
assert npc is mirror
from tenpy_amd.algorithms import module_form  # noqa: E402
from tenpy_amd.linalg import _svd_warm  # noqa: E402


def sync():
    torch.cuda.synchronize()


if args.tebd:
    from tenpy.algorithms import tebd
    from tenpy.models.tf_ising import TFIChain
    from tenpy.networks.mps import MPS
    L = 64
    M = TFIChain({'L': L, 'J': 1., 'g': 1.5, 'bc_MPS': 'finite', 'conserve': 'parity', 'sort_charge': True})
    psi = MPS.from_product_state(M.lat.mps_sites(), ['up'] * L, bc='finite')
    eng = tebd.TEBDEngine(psi, M, {'order': 2, 'dt': 0.05, 'N_steps': 1, 'trunc_params': {'chi_max': args.chi, 'svd_min': 1e-12}})
    log = []
    for k in range(args.sweeps):
        sync()
        t0 = time.time()
        eng.run()
        sync()
        log.append({"step": k, "s": round(time.time() - t0, 3), "max_chi": int(max(psi.chi)),
                    "S_mid": float(psi.entanglement_entropy()[L // 2 - 1])})
    print(json.dumps({"what": "tenpy.algorithms.tebd.TEBDEngine (unmodified) on the device mirror, TFI L=64 g=1.5 parity, order 2, dt=0.05",
                      "steps": log}))
    sys.exit(0)

from tenpy.algorithms import dmrg  # noqa: E402
from tenpy.models.xxz_chain import XXZChain  # noqa: E402
from tenpy.networks.mps import MPS  # noqa: E402
L, chi = args.L, args.chi
M = XXZChain({'L': L, 'Jxx': 1., 'Jz': 1., 'hz': 0., 'bc_MPS': 'finite', 'conserve': 'Sz', 'sort_charge': True})
psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * (L // 2), bc='finite')
opts = {'combine': bool(args.combine), 'max_N_for_ED': 0, 'trunc_params': {'chi_max': min(64, chi), 'svd_min': 1.e-14},
        'lanczos_params': {'N_min': 2, 'N_max': 20}}
if args.mixer:
    opts['mixer'] = True
    opts['mixer_params'] = {'amplitude': 1.e-5, 'decay': 2., 'disable_after': 1000}      # on for the whole ramp; removed below
else:
    opts['mixer'] = None
eng = dmrg.TwoSiteDMRGEngine(psi, M, opts)
log = []


def sweep(tag):
    c0, s0 = mirror.svd_stats['calls'], mirror.svd_stats['sweeps']
    sync()
    t0 = time.time()
    eng.sweep()
    sync()
    log.append({"chi_max": int(eng.trunc_params['chi_max']), "kind": tag, "s": round(time.time() - t0, 3),
                "E": float(eng.update_stats['E_total'][-1]), "max_chi": int(max(psi.chi)),
                "jacobi_sweeps_per_svd": round((mirror.svd_stats['sweeps'] - s0) / max(mirror.svd_stats['calls'] - c0, 1), 2)})
    sys.stderr.write(repr(log[-1]) + "\n")


mix = 'mixer on' if args.mixer else 'no mixer'
c = min(64, chi)
sweep('ramp, ' + mix)
sweep('ramp, ' + mix)
while c < chi:
    c = min(2 * c, chi)
    eng.trunc_params['chi_max'] = c
    if c == chi:
        break
    sweep('ramp, ' + mix)
eng.mixer = None                      # dmrg.py:207-212: the mixer is switched off for the final sweeps
eng.lanczos_params = tenpy.tools.params.asConfig({'N_min': 8, 'N_max': 8}, 'lanczos_params')
for _ in range(args.sweeps):
    sweep('target chi, no mixer, Lanczos N=8')
print(json.dumps({"what": "tenpy.algorithms.dmrg.TwoSiteDMRGEngine (unmodified TeNPy %s) on the device mirror, install(fused=%s), "
                          "Heisenberg L=%d chi=%d, combine=%s" % (tenpy.__version__, bool(args.fused), L, chi, bool(args.combine)),
                  "sweeps": log, "two_site_h": dict(module_form.stats), "svd_warm": dict(_svd_warm.stats),
                  "device": torch.cuda.get_device_name(0)}))
