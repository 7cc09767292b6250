"""Time TeNPy ITSELF (the reference at /root/reference with its compiled ``_npc_helper``, built by ``oracle/build_ref.py``)
on the host cores: SURVEY 8(d) "Timing the CPU path beside it".  Runs in the build container only (the reference cannot
travel to the GPU box); the result is committed as ``profiles/r02_cpu_reference.json`` and quoted by ``bench.py`` as
``cpu_baseline.kind = "reference (offline)"``.

    python scripts/cpu_reference_baseline.py sweep512      # full two-site DMRG sweep, XXZ L=100 chi=512 (BASELINE config 2)
    python scripts/cpu_reference_baseline.py bond2048      # centre-bond updates at chi=2048 (config 3), see below
    python scripts/cpu_reference_baseline.py all

Settings as in the reference's own benchmarks: ``tenpy.tools.optimization.set_level(3)`` (tests/benchmark/tensordot_npc.py:89),
Lanczos ``N_min = N_max = 8`` (fixed work per bond, tests/benchmark/dmrg_infinite.py:36), best of 3
(tests/benchmark/benchmark.py:97), BLAS threads = all cores of this container (count recorded).

``bond2048``: a full chi=2048 sweep would take hours on these cores, so -- as SURVEY 8(d) prescribes -- only centre-bond
updates are timed, on operands that have the charge-block structure of a real chi=2048 Heisenberg state (bond sectors
from the theta dumped on the MI355X, ``scripts/data/theta_chi2048_sat.npz``) and random entries (timing does not depend on
the values with a fixed number of Lanczos steps).  One update = what ``TwoSiteDMRGEngine.update_local`` executes
(algorithms/dmrg.py:529): ``TwoSiteH`` with ``combine=True`` built from LP / RP / W (``combine_Heff``),
``LanczosGroundState.run`` with 8 matvecs, ``svd_theta`` with ``chi_max=2048``, and the environment update ``update_LP``.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402

build_ref.load()
import numpy as np  # noqa: E402
import tenpy  # noqa: E402
import tenpy.linalg.np_conserved as npc  # noqa: E402
from tenpy.algorithms import dmrg  # noqa: E402
from tenpy.algorithms.mps_common import TwoSiteH  # noqa: E402
from tenpy.linalg.krylov_based import LanczosGroundState  # noqa: E402
from tenpy.linalg.truncation import svd_theta  # noqa: E402
from tenpy.models.xxz_chain import XXZChain  # noqa: E402
from tenpy.networks.mpo import MPOEnvironment  # noqa: E402
from tenpy.networks.mps import MPS  # noqa: E402
from tenpy.tools import optimization  # noqa: E402

assert optimization.have_cython_functions, "compiled _npc_helper not active"
optimization.set_level(3)
OUT = os.environ.get('TPA_CPU_REF_OUT') or os.path.join(ROOT, 'profiles', 'r02_cpu_reference.json')


def env_info():
    import scipy
    return {"tenpy": tenpy.__version__, "helper": "compiled _npc_helper (oracle/_ref), without MKL: scipy.linalg.cython_blas",
            "numpy": np.__version__, "scipy": scipy.__version__, "cores": os.cpu_count(),
            "blas_threads": os.environ.get('OMP_NUM_THREADS', 'default (all cores)'), "optimization_level": 3}


def dmrg_protocol(L, chi, n_sweeps_at_chi, label, model='xxz'):
    """bench.py's protocol, sweep by sweep, on the reference: Neel state, 2 sweeps at chi=64, one sweep per doubling of chi
    (adaptive Lanczos N<=20), then `n_sweeps_at_chi` sweeps at the target chi with Lanczos N=8.  Energy and time of EVERY
    sweep are recorded: the energies are what bench.py's `energy_err` compares with.  ``model='hubbard'``: BASELINE config 4,
    FermiHubbardModel on a 2 x (L/2) ladder, t=1, U=8, half filling, charges (N, Sz) (bench.py --config hubbard1024)."""
    if model == 'hubbard':
        from tenpy.models.hubbard import FermiHubbardModel
        M = FermiHubbardModel({'lattice': 'Ladder', 'L': L // 2, 't': 1., 'U': 8., 'mu': 0., 'cons_N': 'N', 'cons_Sz': 'Sz',
                               'bc_MPS': 'finite', 'sort_charge': True})
        psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * (L // 2), bc='finite')
    else:
        M = XXZChain({'L': L, 'Jxx': 1., 'Jz': 1., 'hz': 0., 'bc_MPS': 'finite', 'conserve': 'Sz', 'sort_charge': True})
        psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * (L // 2), bc='finite')
    eng = dmrg.TwoSiteDMRGEngine(psi, M, {'mixer': None, 'combine': True, 'max_N_for_ED': 0,
                                          'trunc_params': {'chi_max': min(64, chi), 'svd_min': 1.e-14},
                                          'lanczos_params': {'N_min': 2, 'N_max': 20}})
    log = []

    def sweep(tag):
        t0 = time.time()
        eng.sweep()
        log.append({"chi_max": int(eng.trunc_params['chi_max']), "lanczos": tag, "s": time.time() - t0,
                    "E": float(eng.update_stats['E_total'][-1]), "max_chi": int(max(psi.chi))})
        print(label, log[-1], flush=True)
    c = min(64, chi)
    sweep('N<=20')
    sweep('N<=20')
    while c < chi:
        c = min(2 * c, chi)
        eng.trunc_params['chi_max'] = c
        if c == chi:
            break
        sweep('N<=20')
    eng.lanczos_params = tenpy.tools.params.asConfig({'N_min': 8, 'N_max': 8}, 'lanczos_params')
    for _ in range(n_sweeps_at_chi):
        sweep('N=8')
    S = psi.get_SL(L // 2)
    at = [e for e in log if e['lanczos'] == 'N=8']
    return {"workload": "bench.py protocol on the reference: %s, TwoSiteDMRGEngine combine=True, no mixer, "
                        "svd_min=1e-14; chi ramp 64 x2, doubling, then %d sweeps at chi=%d with Lanczos N=8"
                        % ("FermiHubbardModel ladder 2 x %d (N, Sz)" % (L // 2) if model == 'hubbard' else "XXZChain L=%d (Sz)" % L,
                           n_sweeps_at_chi, chi),
            "sweeps": log, "s_per_sweep_best": min(e['s'] for e in at[1:]) if len(at) > 1 else at[0]['s'],
            "E_final": at[-1]['E'], "centre_schmidt_values_top16": [float(x) for x in np.sort(S)[::-1][:16]],
            "bond_updates_per_sweep": 2 * (L - 2)}


def sweep512(L=100, chi=512, n_timed=3):
    M = XXZChain({'L': L, 'Jxx': 1., 'Jz': 1., 'hz': 0., 'bc_MPS': 'finite', 'conserve': 'Sz', 'sort_charge': True})
    psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * (L // 2), bc='finite')
    eng = dmrg.TwoSiteDMRGEngine(psi, M, {'mixer': None, 'combine': True, 'max_N_for_ED': 0,
                                          'trunc_params': {'chi_max': 64, 'svd_min': 1.e-14},
                                          'lanczos_params': {'N_min': 2, 'N_max': 20}})
    t0 = time.time()
    c = 64
    eng.sweep()
    eng.sweep()
    while c < chi:
        c = min(2 * c, chi)
        eng.trunc_params['chi_max'] = c
        eng.sweep()
    eng.lanczos_params = tenpy.tools.params.asConfig({'N_min': 8, 'N_max': 8}, 'lanczos_params')
    eng.sweep()                                   # warm-up at the target chi with the timed Lanczos settings
    prep = time.time() - t0
    times = []
    for _ in range(n_timed):
        t0 = time.time()
        eng.sweep()
        times.append(time.time() - t0)
    return {"workload": "TwoSiteDMRGEngine.sweep(), XXZChain L=%d (Sz), chi_max=%d (reached %d), combine=True, no mixer, "
                        "Lanczos N=8, svd_min=1e-14" % (L, chi, max(psi.chi)),
            "s_per_sweep_best": min(times), "s_per_sweep_all": times, "prep_s": prep,
            "E": float(eng.update_stats['E_total'][-1]), "bond_updates_per_sweep": 2 * (L - 2)}


def synthetic_bond(chi_sectors, seed):
    """Operands of one centre-bond update with the given bond sectors (sizes of the charge sectors 2Sz = -9, -7, ..., 9 of
    the bond legs left and right of the two sites; the middle bond follows from them)."""
    L = 100
    M = XXZChain({'L': L, 'Jxx': 1., 'Jz': 1., 'hz': 0., 'bc_MPS': 'finite', 'conserve': 'Sz', 'sort_charge': True})
    i0 = L // 2 - 1
    site = M.lat.mps_sites()[i0]
    ci = site.leg.chinfo
    n = len(chi_sectors)
    q = np.arange(-(n - 1), n, 2)
    vL = npc.LegCharge.from_qind(ci, np.concatenate([[0], np.cumsum(chi_sectors)]), q[:, None], qconj=+1)
    vR = vL.conj()
    W0, W1 = M.H_MPO.get_W(i0), M.H_MPO.get_W(i0 + 1)
    rng = np.random.default_rng(seed)

    def rand(legs, labels):
        return npc.Array.from_func(rng.standard_normal, legs, dtype=np.float64, qtotal=None, shape_kw='size', labels=labels)
    LP = rand([vL, W0.get_leg('wL').conj(), vL.conj()], ['vR*', 'wR', 'vR'])       # 'vR' contracts with theta's 'vL'
    RP = rand([vR.conj(), W1.get_leg('wR').conj(), vR], ['vL', 'wL', 'vL*'])       # 'vL' contracts with theta's 'vR'
    theta = rand([vL, site.leg, site.leg, vR], ['vL', 'p0', 'p1', 'vR'])
    theta /= npc.norm(theta)

    class Env:          # the three things TwoSiteH / _contract_LHeff / _contract_RHeff ask an environment for
        H = M.H_MPO

        @staticmethod
        def get_LP(i, store=True):
            return LP

        @staticmethod
        def get_RP(i, store=True):
            return RP
    Env._contract_LHeff = lambda i, label_p='p0', pipe=None: MPOEnvironment._contract_LHeff(Env, i, label_p, pipe)
    Env._contract_RHeff = lambda i, label_p='p1', pipe=None: MPOEnvironment._contract_RHeff(Env, i, label_p, pipe)
    return Env, i0, theta


def one_update(Env, i0, theta, chi_max):
    t = {}
    t0 = time.time()
    eff = TwoSiteH(Env, i0, combine=True)
    th = eff.combine_theta(theta)
    t['heff'] = time.time() - t0
    t0 = time.time()
    E, th, N = LanczosGroundState(eff, th, {'N_min': 8, 'N_max': 8, 'reortho': False}).run()
    t['lanczos'] = time.time() - t0
    t0 = time.time()
    U, S, VH, err, _ = svd_theta(th, {'chi_max': chi_max, 'svd_min': 1.e-14}, inner_labels=['vR', 'vL'])
    t['svd'] = time.time() - t0
    t0 = time.time()
    U = U.ireplace_label('(vL.p0)', '(vL.p)')
    LPn = npc.tensordot(eff.LHeff, U, axes=['(vR.p0*)', '(vL.p)'])
    LPn = npc.tensordot(U.conj(), LPn, axes=['(vL*.p*)', '(vR*.p0)'])
    t['env'] = time.time() - t0
    t['total'] = sum(t.values())
    t['N_lanczos'] = N
    return t


def bond2048(n_bonds=4):
    import warnings
    d = np.load(os.path.join(ROOT, 'scripts', 'data', 'theta_chi2048_sat.npz'))
    rows = sorted([int(d[k].shape[0]) for k in d.files])
    # fused (vL.p0) sector sizes are sums of neighbouring bond sectors; a symmetric set of 10 bond sectors with the same
    # total (2048) and the same large fused blocks (1084, 876, 876, 456, 456, ...) as the dump (1072, 872, 870, 461, 456, ...)
    sectors = [2, 24, 122, 334, 542, 542, 334, 122, 24, 2]
    res = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for b in range(n_bonds):
            Env, i0, theta = synthetic_bond(sectors, seed=b)
            res.append(one_update(Env, i0, theta, 2048))
            print("bond", b, res[-1], flush=True)
    best = min(res, key=lambda t: t['total'])
    return {"workload": "centre-bond update of two-site DMRG at chi=2048 on operands with the block structure of a chi=2048 "
                        "Heisenberg state (bond sectors %r, random entries): TwoSiteH(combine=True) + LanczosGroundState "
                        "(8 matvecs) + svd_theta(chi_max=2048) + update_LP" % (sectors,),
            "fused_rows_of_the_dumped_theta": rows, "s_per_bond_best": best['total'], "best": best, "all": res,
            "s_per_sweep_extrapolated": best['total'] * 196,
            "extrapolation": "x196 bond updates; upper bound: the ~20 bonds next to each edge are smaller"}


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    out = json.load(open(OUT)) if os.path.exists(OUT) else {}
    out['environment'] = env_info()
    if what in ('bond2048', 'all'):
        out['bond2048'] = bond2048(int(os.environ.get('N_BONDS', 4)))
        json.dump(out, open(OUT, 'w'), indent=1)
    if what.startswith('dmrg'):            # e.g. dmrg2048:3  or  dmrg512:4
        chi, n = (what[4:].split(':') + ['3'])[:2]
        out['dmrg%s' % chi] = dmrg_protocol(100, int(chi), int(n), 'dmrg%s' % chi)
        json.dump(out, open(OUT, 'w'), indent=1)
    if what.startswith('hubbard'):         # e.g. hubbard1024:3   (ladder 2 x 40)
        chi, n = (what[7:].split(':') + ['3'])[:2]
        out['hubbard%s' % chi] = dmrg_protocol(80, int(chi), int(n), 'hubbard%s' % chi, model='hubbard')
        json.dump(out, open(OUT, 'w'), indent=1)
    if what in ('sweep512', 'all'):
        out['sweep512'] = sweep512()
        json.dump(out, open(OUT, 'w'), indent=1)
    print(json.dumps(out, indent=1))
