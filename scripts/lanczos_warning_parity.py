"""How often does the forced-N Lanczos protocol of the bench (N_min = N_max = 8, run past convergence) produce "poorly conditioned H
matrix in KrylovBased" (reference krylov_based.py:183-188) -- in TeNPy's OWN engine on the host and in the backend's native loop (which
evaluates the reference's stopping test one step late, tenpy_amd/linalg/krylov_based.py)?  Same model, chi, sweeps, start state;
VERDICT r5 ("nobody has shown that TeNPy's own _build_krylov warns on the same bonds of the same protocol").
    python scripts/lanczos_warning_parity.py [L] [chi] [n_sweeps]        (CPU: the backend runs on the numpy emulation of the device)"""
import json
import logging
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
chi = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n_sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 10


class Count(logging.Handler):
    def __init__(self):
        super().__init__()
        self.n = 0
        self.vals = []

    def emit(self, record):
        msg = record.getMessage()
        if 'poorly conditioned' in msg:
            self.n += 1
            self.vals.append(msg.split('=')[-1].strip())


def run_reference(diag_method='default'):
    from oracle import build_ref
    build_ref.load()
    import tenpy
    from tenpy.algorithms import dmrg
    from tenpy.models.xxz_chain import XXZChain
    from tenpy.networks.mps import MPS
    h = Count()
    logging.getLogger('tenpy.linalg.krylov_based').addHandler(h)
    logging.getLogger('tenpy.linalg.krylov_based').setLevel(logging.WARNING)
    M = XXZChain({'L': L, 'Jxx': 1., 'Jz': 1., 'hz': 0., 'bc_MPS': 'finite', 'conserve': 'Sz', 'sort_charge': True})
    psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * (L // 2), bc='finite')
    eng = dmrg.TwoSiteDMRGEngine(psi, M, {'trunc_params': {'chi_max': chi, 'svd_min': 1e-14}, 'mixer': None, 'combine': True,
                                          'diag_method': diag_method, 'lanczos_params': {'N_min': 8, 'N_max': 8}})
    per_sweep, E = [], []
    for _ in range(n_sweeps):
        n0 = h.n
        eng.sweep()
        per_sweep.append(h.n - n0)
        E.append(float(eng.sweep_stats['E'][-1]) if eng.sweep_stats.get('E') else None)
    return {"engine": "TeNPy %s TwoSiteDMRGEngine, diag_method=%r (default: full_diag_effH below eff_H.N = 400, dmrg.py:733-739; 'lanczos': "
                      "LanczosGroundState on every bond)" % (tenpy.__version__, diag_method), "warnings_per_sweep": per_sweep,
            "E": float(sum(psi.expectation_value_term([('Sz', i), ('Sz', i + 1)]) for i in range(0))) if False else None}


def run_backend():
    class _MP:
        def setattr(self, o, n, v):
            setattr(o, n, v)
    import mock_device
    mock_device.install(_MP())
    from tenpy_amd.algorithms.dmrg import TwoSiteDMRGEngine
    from tenpy_amd.linalg import krylov_based as kb
    from tenpy_amd.models.spin_chains import spin_half_leg, xxz_chain_mpo
    from tenpy_amd.networks.mps import MPS
    h = Count()
    logging.getLogger(kb.logger.name).addHandler(h)
    H = xxz_chain_mpo(L, 1., 1., 0.)
    _, p = spin_half_leg('Sz')
    psi = MPS.from_product_state([p] * L, [1, 0] * (L // 2))
    eng = TwoSiteDMRGEngine(psi, H, {'trunc_params': {'chi_max': chi, 'svd_min': 1e-14}, 'lanczos_params': {'N_min': 8, 'N_max': 8}})
    per_sweep = []
    for _ in range(n_sweeps):
        n0 = kb.stats['n_ill_conditioned']
        eng.sweep()
        per_sweep.append(kb.stats['n_ill_conditioned'] - n0)
    return {"engine": "tenpy_amd stand-alone driver, native Lanczos loop (numpy emulation of the device)", "warnings_per_sweep": per_sweep,
            "n_degenerate": kb.stats['n_degenerate'], "E": float(eng.sweep_stats['E'][-1])}


if __name__ == '__main__':
    which = os.environ.get('WHICH')
    if which == 'ref':
        print(json.dumps(run_reference()))
    elif which == 'ref_lanczos':
        print(json.dumps(run_reference('lanczos')))
    elif which == 'backend':
        print(json.dumps(run_backend()))
    else:
        import subprocess
        out = {"protocol": "XXZ (Heisenberg) L=%d, Neel start, two-site DMRG chi_max=%d svd_min=1e-14, Lanczos N_min=N_max=8, %d sweeps, bond updates per sweep %d"
                           % (L, chi, n_sweeps, 2 * (L - 2))}
        for w in ('ref', 'ref_lanczos', 'backend'):
            pr = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(os.environ, WHICH=w), capture_output=True, text=True)
            line = [l for l in pr.stdout.splitlines() if l.startswith('{')]
            out[w] = json.loads(line[-1]) if line else {"error": pr.stderr[-2000:]}
        print(json.dumps(out, indent=1))
