#!/usr/bin/env python
"""bench.py -- two-site DMRG sweep time (+ energy / singular-value error) on MI355X: BASELINE.json's metric.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config heis2048|xxz512|hubbard1024|tebd1024] [--chi CHI] [--L L]

Default workload (``config["workload"]``): spin-1/2 Heisenberg chain (XXZ, Jxx=Jz=1, hz=0, Sz conserved), L=100, two-site
DMRG at bond dimension chi=2048 (BASELINE.json's headline configuration; fits one GPU), Lanczos with N_min=N_max=8 (fixed
work per bond, as the reference's tests/benchmark/dmrg_infinite.py:36 does), svd_min=1e-14, no mixer.  A "step" is ONE FULL
SWEEP = 2(L-2) = 196 two-site bond updates (effective-H build, 8-step Lanczos, block SVD + truncation, environment update).
The other BASELINE configurations are selected with ``--config`` (same JSON shape): ``xxz512`` (config 2), ``hubbard1024``
(config 4: Fermi-Hubbard ladder 2x40, U(1)xU(1)), ``tebd1024`` (config 5: TFI L=64, parity, complex128 real-time TEBD of
order 2; a step is one ``evolve_step`` over all bonds of a random chi=1024 state, i.e. at saturated bond dimension).

The MPS is synthetic in the sense of the contract: there is no checkpoint to load, so the state is grown on the device from
the Neel product state by an (untimed) chi ramp of single sweeps chi = 64, 64, 128, ..., chi/2, followed by W warm-up
sweeps at the target chi; then exactly K sweeps are timed between barrier + torch.cuda.synchronize() on both sides.

One JSON line on rank 0:
* ``roofline`` describes the DOMINANT kernel family of the timed region, the batched block SVD (``tpa_svd_batch``): HIP
  events on the launch stream around every call, algorithmic flops ``4 m^2 n + 8 m n^2 + 9 n^3`` and bytes
  ``8 (mn + mk + kn + k)`` per charge block, against the fp64 MFMA peak and (``hbm_frac``) the HBM peak;
  ``roofline_gemm`` is the same for the grouped MFMA GEMM (tensordot / Lanczos matvec / environment update).
* parity at scale (N=1): ``sv_max_rel_err`` (block SVD of the centre-bond theta vs LAPACK through the numpy oracle,
  max |dS| / S_max), ``matvec_max_rel_err``, ``E0_rel_err`` (8-step Lanczos energy of the centre bond vs the oracle's
  Lanczos on the same operator) and the top-level ``energy_err`` = |E - E_ref| / |E_ref| against the energy the REFERENCE
  (TeNPy with its compiled helper, run offline in the build container with the same protocol) reaches after the same
  sweep (``profiles/r02_cpu_reference.json``); ``null`` when that file has no entry for this configuration / sweep.
* ``cpu_baseline``: kind "reference (offline, 8 cores)" = TeNPy itself, measured in the build container
  (``scripts/cpu_reference_baseline.py``; it cannot travel to the GPU box), with the numpy oracle timed live on this
  host's cores on a bounded sample as the secondary field ``port``.

Multi-GPU (--gpus N, launched by torch.distributed.run, one rank per GPU over RCCL): the Lanczos matvec is sharded over the
ranks by rows of theta' (tenpy_amd/algorithms/sharded.py) and the charge blocks of every SVD are distributed (DESIGN.md
section 5).  Total work is fixed -> "scaling": "strong"; the value is the max over ranks of the time per sweep.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6     # MI355X fp64 matrix peak (vendor figure); not in the microarch guide
PEAK_HBM_GBS = 8000.0
CPU_REF = os.path.join(ROOT, 'profiles', 'r02_cpu_reference.json')


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', default=os.environ.get('TPA_BENCH_CONFIG', 'heis2048'),
                    choices=['heis2048', 'xxz512', 'hubbard1024', 'tebd1024'])
    ap.add_argument('--chi', type=int, default=int(os.environ.get('TPA_BENCH_CHI', 0)))
    ap.add_argument('--L', type=int, default=0)
    ap.add_argument('--lanczos-N', type=int, default=8)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample-bonds', type=int, default=3)
    ap.add_argument('--qr', action='store_true', help="tebd1024: QR-based truncation (decompose_theta_qr_based, reference "
                    "algorithms/tebd.py:685) instead of the block SVD of theta")
    ap.add_argument('--eig-svd', action='store_true', help="with --qr: _eig_based_svd for the bond matrix (truncation.py:473)")
    ap.add_argument('--force-dist', action='store_true', help="world size 1 only: still initialise torch.distributed (RCCL on the GPU) and run the "
                    "row-sharded engine, so that the collective of the N > 1 path executes on a one-GPU box (TPA_BENCH_FORCE_DIST=1 does the same)")
    ap.add_argument('--no-extras', action='store_true', help="heis2048 on one GPU: skip the legs after the headline (adaptive Lanczos sweep, "
                    "other BASELINE configurations, TeNPy's own engine on the device, vector-kernel roofline)")
    return ap.parse_args(argv)


CONFIGS = {     # name: (default L, default chi, description)
    'heis2048': (100, 2048, "spin-1/2 Heisenberg chain (XXZ Jxx=Jz=1, Sz conserved)"),
    'xxz512': (100, 512, "spin-1/2 XXZ chain Jxx=Jz=1 (Sz conserved)"),
    'hubbard1024': (80, 1024, "Fermi-Hubbard ladder 2x40, t=1, U=8, half filling, charges (N, 2Sz)"),
    'tebd1024': (64, 1024, "TFI chain J=1, g=1.5 (parity conserved), real-time TEBD order 2, dt=0.05, complex128"),
}


def oracle_tensor(arr):
    """Device Array -> oracle OTensor (host copies of the blocks)."""
    from oracle import npc_oracle as orc
    legs = [orc.OLeg(l.slices, l.charges, l.qconj, arr.chinfo.mod) for l in arr.legs]
    return orc.OTensor(legs, arr.qtotal, arr._qdata, arr._data)


def _legs_blocked(t):
    return all(len({tuple(c) for c in l.charges.tolist()}) == len(l.charges) for l in t.legs)


def parity_and_port(eng, args, gpu_bond_s):
    """On `cpu_sample_bonds` centre bonds of the SAME state: run the numpy oracle (8 effective-H matvecs + block SVD; its
    time is the `port` baseline) and compare the device results with it.  Returns (port dict, parity dict)."""
    from oracle import npc_oracle as orc
    from tenpy_amd.algorithms.mps_common import TwoSiteH
    from tenpy_amd.linalg import np_conserved as npc
    from tenpy_amd.linalg.krylov_based import LanczosGroundState
    L = eng.psi.L
    n_b = max(1, args.cpu_sample_bonds)
    # an edge bond (tiny charge sectors: the Lanczos input is embedded with zero blocks, `TwoSiteH.native_input`), a quarter bond
    # and the centre (two centre bonds only); with fewer samples the centre first
    spread = [L // 2 - 1, 2, L // 4, L // 2, 3 * L // 4]
    bonds = spread[:n_b] if n_b <= len(spread) else [L // 2 - 1 + i for i in range(n_b)]
    t_cpu, n_centre, mv_err, sv_err, e0_err, sv_ind, iso = 0., 0, [], [], [], [], []
    hp = None
    n_kept = n_kept_bad = n_kept_abs_bad = 0
    for i0 in bonds:
        eff = TwoSiteH(eng.env, i0, factored=False)       # the fused form LHeff . theta . RHeff that the oracle restates
        theta = eff.combine_theta(eng.psi.get_theta(i0, n=2))
        fac = TwoSiteH(eng.env, i0)                       # what the timed sweeps ran (factored when W has scalar blocks)
        th_fac = fac.combine_theta(eng.psi.get_theta(i0, n=2))
        want = fac.prepare_svd(fac.matvec(th_fac))
        LH, RH, th = oracle_tensor(eff.LHeff), oracle_tensor(eff.RHeff), oracle_tensor(theta)
        # --- device side of the comparison
        if os.environ.get('TPA_DUMP_THETA') and i0 == bonds[0]:
            # charge blocks of the wave function the SVD sees (after one Lanczos run), for scripts/svd_file_bench.py: this is how
            # scripts/data/theta_chi2048_sat.npz (not tracked: 25 MB) is produced -- `TPA_DUMP_THETA=path python bench.py`
            _, th_opt, _ = LanczosGroundState(fac, th_fac, {'N_min': args.lanczos_N, 'N_max': args.lanczos_N}).run()
            np.savez(os.environ['TPA_DUMP_THETA'], **{'b%02d' % k: np.asarray(b) for k, b in enumerate(fac.prepare_svd(th_opt)._data)})
        npc.svd_engine_floor = True       # the parity sample measures the SVD as the timed sweeps ran it (the engines' floor + clean-up)
        U, S_dev, VH = npc.svd(fac.prepare_svd(th_fac), inner_labels=['vR', 'vL'])
        E_dev, _, N_dev = LanczosGroundState(fac, th_fac, {'N_min': args.lanczos_N, 'N_max': args.lanczos_N}).run()
        # --- oracle: timed part = N matvecs (+ the vector work of a Lanczos step) + block SVD
        t0 = time.time()
        E_orc, _, N_orc = orc.lanczos_gs(lambda v: orc.matvec_two_site(LH, RH, v), th, N_min=args.lanczos_N, N_max=args.lanczos_N)
        blocked = th if _legs_blocked(th) else orc.combine_legs(th, [[0], [1]], [th.legs[0].qconj, th.legs[1].qconj])[0]
        _, S_orc, _ = orc.svd(blocked)
        if abs(i0 - (L // 2 - 1)) <= 1:       # the `port` baseline extrapolates from the centre bond(s) only (the edge bond costs nothing)
            t_cpu += time.time() - t0
            n_centre += 1
        first = orc.matvec_two_site(LH, RH, th).to_dense()
        mv_err.append(float(np.max(np.abs(want.to_ndarray() - first)) / max(np.max(np.abs(first)), 1e-300)))
        a, b = np.sort(np.asarray(S_dev))[::-1], np.sort(np.asarray(S_orc))[::-1]
        n = min(len(a), len(b))
        sv_err.append(float(np.max(np.abs(a[:n] - b[:n])) / b[0]))
        big = b[:n] > 1.e-8 * b[0]
        sv_ind.append(float(np.max(np.abs(a[:n][big] - b[:n][big]) / b[:n][big])))
        # what DMRG keeps of this bond: the chi_max largest values above svd_min (relative to the norm) -- how many of THOSE are off by
        # more than 1e-10 of their own size (makes the "relative to sigma_max" reading of north_star's 1e-10 checkable)
        keep_n = min(n, int(eng.trunc_params.get('chi_max', n)))
        kept = b[:keep_n] > float(eng.trunc_params.get('svd_min', 0.)) * np.linalg.norm(b)
        n_kept += int(kept.sum())
        n_kept_bad += int(np.sum(np.abs(a[:keep_n][kept] - b[:keep_n][kept]) > 1.e-10 * b[:keep_n][kept]))
        # ... and by more than LAPACK's own error bound for a singular value, a few eps sigma_max (the ORACLE is LAPACK: below
        # ~1e-6 sigma_max its values are not good to 1e-10 of their own size either; one-sided Jacobi is the more accurate of the two there)
        n_kept_abs_bad += int(np.sum(np.abs(a[:keep_n][kept] - b[:keep_n][kept]) > 32 * np.finfo(float).eps * b[0]))
        # isometry defect over all kept vectors (sigma > 1e-14 sigma_max, what svd_min = 1e-14 keeps)
        Ud, Vd = U.to_ndarray(), VH.to_ndarray()
        kept = np.asarray(S_dev) > 1.e-14 * np.max(S_dev)
        iso.append(float(max(np.max(np.abs(Ud[:, kept].conj().T @ Ud[:, kept] - np.eye(int(kept.sum())))),
                             np.max(np.abs(Vd[kept] @ Vd[kept].conj().T - np.eye(int(kept.sum())))))))
        e0_err.append(abs(E_dev - E_orc) / abs(E_orc))
        if i0 == bonds[0] and hp is None:
            hp = _sv_vs_highprec(fac.prepare_svd(th_fac), np.asarray(S_dev))
    # ---- what the timed sweeps themselves produced (whatever path each bond's SVD took: warm, sketch, cold): isometry of the stored
    #      MPS tensors, sum_{vL, p} conj(A) A = 1 (form A) or sum_{p, vR} B conj(B) = 1 (form B), at the sampled bonds' sites
    mps_iso = []
    for i in sorted({min(max(b + d, 1), L - 2) for b in bonds for d in (0, 1)}):
        T = eng.psi.get_B(i, None)
        f = tuple(eng.psi.form[i])
        if f == (1., 0.):
            G = npc.tensordot(T.conj(), T, axes=(['vL*', 'p*'], ['vL', 'p'])).to_ndarray()
        elif f == (0., 1.):
            G = npc.tensordot(T, T.conj(), axes=(['p', 'vR'], ['p*', 'vR*'])).to_ndarray()
        else:
            continue
        mps_iso.append(float(np.max(np.abs(G - np.eye(G.shape[0])))))
    per_bond = t_cpu / max(n_centre, 1)
    n_bonds = 2 * (L - 2)
    port = {"value": per_bond * n_bonds, "unit": "s/sweep", "cores": os.cpu_count(), "kind": "port",
            "sample": "%d centre bond update(s) (%d-step Lanczos + block SVD each) with the numpy oracle on the same state, "
                      "%.2f s CPU per bond, extrapolated x%d bonds; GPU same bonds %.4f s per bond"
                      % (max(n_centre, 1), args.lanczos_N, per_bond, n_bonds, gpu_bond_s)}
    parity = {"sv_max_rel_err": max(sv_err), "sv_max_rel_err_individual": max(sv_ind), "sv_kept": n_kept,
              "sv_kept_rel_err_over_1e-10": n_kept_bad, "sv_kept_abs_err_over_32eps_smax": n_kept_abs_bad, "svd_isometry_defect": max(iso),
              "mps_isometry_defect": max(mps_iso) if mps_iso else None,
              "matvec_max_rel_err": max(mv_err), "E0_rel_err": max(e0_err), "sv_vs_highprec": hp,
              "parity_sample": "bonds %r (centre, edge, quarter) of the timed state: device block SVD vs LAPACK (oracle; sv_max_rel_err = max |dS| / S_max, "
                               "..._individual = max |dS_i| / S_i over S_i > 1e-8 S_max, mps_isometry_defect = max |T^H T - 1| of the MPS tensors the timed sweeps stored at "
                               "those sites (whatever path their SVDs took), svd_isometry_defect = max(|U^H U - 1|, |VH VH^H - 1|) "
                               "over the vectors with S > 1e-14 S_max), factored device matvec "
                               "vs oracle LHeff.theta.RHeff, %d-step Lanczos energy vs the oracle's Lanczos" % (bonds, args.lanczos_N)}
    return port, parity


def _sv_vs_highprec(theta, S_dev, rows=(100, 330)):
    """the device singular values of ONE mid-size charge block of the centre-bond wave function against singular values
    in EXTENDED precision (tests/svd_reference.py: np.longdouble, checked against mpmath in tests/test_svd_highprec.py), next to
    LAPACK's on the same block -- LAPACK cannot arbitrate values below ~1e-6 sigma_max, its own relative error there is 1e-10 ... 1e-3.
    Returns the worst relative error over the decades >= 1e-12 sigma_max for both, and per decade."""
    try:
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        import svd_reference as sr
        blocks = theta._data
        shapes = [b.shape for b in blocks]
        ks = [min(sh) for sh in shapes]
        cand = [b for b in range(len(blocks)) if rows[0] <= ks[b] <= rows[1]]
        if not cand:
            return None
        b = max(cand, key=lambda x: ks[x])
        off = int(sum(ks[:b]))
        A = np.asarray(blocks[b], dtype=np.float64)
        ref = sr.sv_reference(A)
        e_dev = sr.rel_err_by_decade(S_dev[off:off + ks[b]], ref)
        e_lap = sr.rel_err_by_decade(np.linalg.svd(A, compute_uv=False), ref)
        dec = [d for d in e_dev if d <= 12 and d in e_lap]
        return {"block": list(shapes[b]), "sv_max_rel_err_vs_highprec": max(e_dev[d] for d in dec), "lapack_max_rel_err_vs_highprec": max(e_lap[d] for d in dec),
                "worst_ratio_to_lapack": max(e_dev[d] / max(e_lap[d], 1e-300) for d in dec),
                "by_decade": {str(d): [float("%.2g" % e_dev[d]), float("%.2g" % e_lap[d])] for d in sorted(e_dev) if d in e_lap},
                "note": "max relative error per decade of sigma / sigma_max, [device, LAPACK], against np.longdouble singular values; decades 0..12 in the maxima"}
    except Exception as e:      # a checker: never kills the line
        return {"error": repr(e)}


def reference_same_run(n_bonds=2, timeout=600):
    """TeNPy ITSELF (the reference, compiled helper) timed in THIS run on this host's cores: centre-bond updates at chi = 2048
    (scripts/cpu_reference_baseline.py bond2048: TwoSiteH + 8-step Lanczos + svd_theta + update_LP on operands with the block
    structure of the real state), x196 bonds.  The reference comes from /root/reference or, on the GPU box, from the archive
    oracle/_ref/tenpy_ref.zip (oracle/build_ref.py).  None if it is not available.  Bounded: ~2 bonds x ~4 s x cores-dependent."""
    import subprocess
    import tempfile
    try:
        from oracle import build_ref
        if build_ref.reference_root() is None or not os.path.exists(build_ref.SO):
            return None
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, 'cpu_ref.json')
            env = dict(os.environ, TPA_CPU_REF_OUT=out, N_BONDS=str(n_bonds))
            subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'cpu_reference_baseline.py'), 'bond2048'], env=env,
                           capture_output=True, text=True, timeout=timeout, check=True)
            with open(out) as f:
                r = json.load(f)
        b, envi = r['bond2048'], r['environment']
        return {"value": b['s_per_sweep_extrapolated'], "unit": "s/sweep", "cores": envi.get('cores'), "kind": "reference",
                "where": "same run: host cores of the GPU box (%s cores, %s BLAS threads)" % (envi.get('cores'), envi.get('blas_threads')),
                "sample": "TeNPy %s with its compiled _npc_helper: best of %d centre-bond updates at chi=2048 (%.2f s each: Lanczos "
                          "%.2f, svd_theta %.2f, H_eff %.2f, env %.2f) x 196 bonds" % (
                              envi.get('tenpy'), n_bonds, b['s_per_bond_best'], b['best']['lanczos'], b['best']['svd'],
                              b['best']['heff'], b['best']['env'])}
    except Exception as e:      # the baseline must never kill the bench line
        sys.stderr.write("reference_same_run failed: %r\n" % (e,))
        return None


def reference_entry(config_name, chi):
    """Offline numbers of TeNPy itself for this configuration (profiles/r02_cpu_reference.json), or None."""
    try:
        with open(CPU_REF) as f:
            ref = json.load(f)
    except Exception:
        return None, None
    if config_name == 'hubbard1024':
        return ref.get('hubbard%d' % chi), ref
    return ref.get('dmrg%d' % chi) if config_name in ('heis2048', 'xxz512') else None, ref


def build_dmrg(args, world, name):
    from tenpy_amd.algorithms.dmrg import TwoSiteDMRGEngine
    from tenpy_amd.networks.mps import MPS
    L, chi = args.L, args.chi
    if name == 'hubbard1024':
        from tenpy_amd.models.hubbard import hubbard_ladder_mpo, spinful_fermion_leg
        H = hubbard_ladder_mpo(L // 2, 1., 8., 0.)
        _, p = spinful_fermion_leg()
        psi = MPS.from_product_state([p] * L, [1, 2] * (L // 2))
    else:
        from tenpy_amd.models.spin_chains import spin_half_leg, xxz_chain_mpo
        H = xxz_chain_mpo(L, 1., 1., 0.)
        _, p = spin_half_leg('Sz')
        psi = MPS.from_product_state([p] * L, [1, 0] * (L // 2))
    eng = TwoSiteDMRGEngine(psi, H, {'trunc_params': {'chi_max': min(64, chi), 'svd_min': 1.e-14},
                                     'lanczos_params': {'N_min': 2, 'N_max': 20}, 'shard_matvec': world > 1,
                                     'profile': bool(os.environ.get('TPA_BENCH_PHASES'))})
    # ---- untimed: grow the state (two quick sweeps at small chi, then double chi per sweep)
    energies = []
    eng.ramp_log = []       # (chi_max, seconds, Jacobi sweeps per SVD call) of every untimed sweep: the non-steady-state regime

    def timed_sweep(tag):
        import torch
        from tenpy_amd.linalg import np_conserved as npc
        c0, s0 = npc.svd_stats['calls'], npc.svd_stats['sweeps']
        torch.cuda.synchronize()
        t0 = time.time()
        eng.sweep()
        torch.cuda.synchronize()
        eng.ramp_log.append({"chi_max": int(eng.trunc_params['chi_max']), "s": round(time.time() - t0, 3), "kind": tag,
                             "jacobi_sweeps_per_call": round((npc.svd_stats['sweeps'] - s0) / max(npc.svd_stats['calls'] - c0, 1), 2)})
        energies.append(eng.sweep_stats['E'][-1])
    eng.timed_sweep = timed_sweep
    c = min(64, chi)
    for _ in range(2):
        timed_sweep('ramp, Lanczos N<=20')
    while c < chi:
        c = min(2 * c, chi)
        eng.trunc_params['chi_max'] = c
        if c == chi:
            break
        timed_sweep('ramp, Lanczos N<=20')
    eng.lanczos_params = {'N_min': args.lanczos_N, 'N_max': args.lanczos_N}
    return eng, energies


def build_tebd(args):
    """Random right-canonical MPS at the full bond dimension (two parity sectors of chi/2 each, complex128) so that the
    timed evolve_step runs at saturated chi from the first step, as BASELINE config 5 specifies."""
    from tenpy_amd.algorithms.tebd import QRBasedTEBDEngine, TEBDEngine
    from tenpy_amd.models.spin_chains import spin_half_leg
    from tenpy_amd.networks.mps import MPS
    L, chi = args.L, args.chi
    J, g = 1., 1.5
    _, p = spin_half_leg('parity')
    sx, sz, I2 = np.array([[0., 1.], [1., 0.]]), np.diag([-1., 1.]), np.eye(2)
    h_bonds = [None]
    for i in range(1, L):
        gl = g if i - 1 == 0 else g / 2
        gr = g if i == L - 1 else g / 2
        h_bonds.append((-J * np.kron(sx, sx) - gl * np.kron(sz, I2) - gr * np.kron(I2, sz)).reshape(2, 2, 2, 2))
    psi = random_right_canonical_mps(p, L, chi, np.complex128, seed=1)
    if args.qr:     # the reference's GPU-motivated route -- QR of theta Y0 instead of the SVD of theta
        return QRBasedTEBDEngine(psi, h_bonds, {'dt': 0.05, 'compute_err': True, 'cbe_expand': 0.1, 'use_eig_based_svd': bool(args.eig_svd),
                                                'trunc_params': {'chi_max': chi, 'svd_min': 1e-12}})
    opts = {'dt': 0.05, 'compute_err': False, 'trunc_params': {'chi_max': chi, 'svd_min': 1e-12}}
    if int(os.environ.get('WORLD_SIZE', '1')) > 1 and not os.environ.get('TPA_TEBD_REPLICAS'):
        # N > 1: the bonds of every half-step dealt over the ranks, new tensors broadcast from their owners (algorithms/sharded.py)
        from tenpy_amd.algorithms.sharded import ShardedTEBDEngine
        return ShardedTEBDEngine(psi, h_bonds, opts)
    if os.environ.get('TPA_TEBD_BATCH'):          # measurement knob: k bonds of a half-step per batched SVD call (tebd.py: update_bonds_batched)
        opts['batch_bonds'] = int(os.environ['TPA_TEBD_BATCH'])
    return TEBDEngine(psi, h_bonds, opts)


def random_right_canonical_mps(p, L, chi, dtype, seed):
    """Random MPS in right-canonical ('B') form (scripts/tebd_state.py: the generator is shared with the offline TeNPy run of the same
    quench, scripts/cpu_reference_tebd.py, so that both start from the same state)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'scripts'))
    import tebd_state
    from tenpy_amd.linalg import np_conserved as npc
    from tenpy_amd.linalg.charges import LegCharge, LegPipe
    from tenpy_amd.networks.mps import MPS
    Bs, Ss = tebd_state.random_right_canonical_tensors(npc, LegCharge, LegPipe, p, L, chi, dtype, seed)
    return MPS([p] * L, Bs, Ss, form='B')


def run(argv=None, emit=True):
    """One bench line (dict); printed as JSON by rank 0 when ``emit``."""
    args = parse(argv)
    out = None
    L0, chi0, desc = CONFIGS[args.config]
    args.L = args.L or L0
    args.chi = args.chi or chi0
    import torch
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    backend = os.environ.get('TPA_BENCH_BACKEND', 'nccl')     # 'gloo': CPU dry run of the N>1 control flow (tests/test_bench_contract.py)
    force_dist = world == 1 and (args.force_dist or bool(os.environ.get('TPA_BENCH_FORCE_DIST')))
    if world > 1 or force_dist:
        import torch.distributed as dist
        if force_dist:              # a one-rank group: the collectives of the N > 1 path run (RCCL on the GPU) with nobody to talk to
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', str(_free_port()))
            os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', TPA_SHARD_FORCE='1')
        if backend == 'nccl':
            torch.cuda.set_device(local)
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend=backend)
    else:
        dist = None
        torch.cuda.set_device(0)

    from tenpy_amd.linalg import np_conserved as npc
    from tenpy_amd.linalg._device import lib as dev_lib
    L, chi = args.L, args.chi
    is_tebd = args.config == 'tebd1024'
    if os.environ.get('TPA_QR_ALG'):         # measurement knob: 2 = the two-launches-per-panel QR of rounds 2-4 (tpa_qr_set_algorithm)
        dev_lib().tpa_qr_set_algorithm(int(os.environ['TPA_QR_ALG']))
    t_prep = time.time()
    if is_tebd:
        eng, ramp_E = build_tebd(args), []
        step = eng.evolve_step_order2
    else:
        eng, ramp_E = build_dmrg(args, 2 if force_dist else world, args.config)
        step = eng.sweep
    sweep_E = list(ramp_E)
    n_ramp = len(ramp_E)
    for _ in range(args.warmup):
        if is_tebd:
            step()
        else:
            eng.timed_sweep('warm-up at the target chi, Lanczos N=%d' % args.lanczos_N)
    if not is_tebd:
        sweep_E = list(ramp_E)      # timed_sweep appends to the same list
    torch.cuda.synchronize()
    t_prep = time.time() - t_prep

    if not is_tebd:
        eng.phase_time = {k: 0. for k in eng.phase_time}
    # ---- timed region: exactly K steps
    for tm in (npc.gemm_timer, npc.svd_timer, npc.eigh_timer):
        tm.keep = bool(os.environ.get('TPA_BENCH_SVD_RECORDS')) and tm is npc.svd_timer
        tm.reset()
        tm.enabled = True
    # the grouped-GEMM rate is measured on every 8th launch (Lanczos run): an event record between two kernels is a barrier packet of
    # ~5.6 us, and four of them per matvec were 1.7 % of the sweep (KernelTimer.sample); the SVD / eigh timers bracket whole calls
    npc.gemm_timer.stride = int(os.environ.get('TPA_BENCH_GEMM_TIMER_STRIDE', '8'))
    n0 = 0 if is_tebd else len(eng.update_stats['E_total'])
    from tenpy_amd.linalg import _svd_warm as _sw
    npc.svd_stats['max_block'] = 0          # (a running maximum: the other configurations of the extras run in this process too)
    svd_calls0, svd_sweeps0, warm0 = npc.svd_stats['calls'], npc.svd_stats['sweeps'], dict(_sw.stats)
    import ctypes as _ct
    from tenpy_amd.linalg import krylov_based as _kb
    lanczos0 = dict(_kb.stats)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        step()
        if not is_tebd:
            sweep_E.append(eng.sweep_stats['E'][-1])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.time() - t0
    for tm in (npc.gemm_timer, npc.svd_timer, npc.eigh_timer):
        tm.enabled = False
        tm.collect()
    if os.environ.get('TPA_BENCH_SVD_RECORDS'):       # diagnostic: HIP-event time, kind, sweeps and largest block of every timed SVD call
        with open(os.environ['TPA_BENCH_SVD_RECORDS'], 'w') as f:
            json.dump([[t] + list(tag or ()) for t, tag in npc.svd_timer.records], f)
        _log = (_ct.c_int64 * (8 * 8192))()
        _n = dev_lib().tpa_svd_call_log(_log, 8192, 0)
        with open(os.environ['TPA_BENCH_SVD_RECORDS'] + '.calls', 'w') as f:       # the library's own log of its last tpa_svd_batch calls
            json.dump([list(_log[8 * i:8 * i + 8]) for i in range(_n)], f)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    s_per_step = elapsed / max(args.steps, 1)

    if rank == 0:
        def roof(tm, kernel, extra):
            sec = tm.ms * 1e-3
            tflops = tm.flops / sec / 1e12 if sec > 0 else 0.
            gbs = tm.bytes_min / sec / 1e9 if sec > 0 else 0.
            r = {"bound": "mfma", "achieved": tflops, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                 "frac": tflops / PEAK_FP64_MFMA_TFLOPS, "traffic": None,
                 "hbm_achieved_GBs": gbs, "hbm_frac": gbs / PEAK_HBM_GBS, "kernel": kernel, "launches": tm.n_launch,
                 "avg_launch_ms": tm.ms / max(tm.n_launch, 1), "algorithmic_flops_per_launch": tm.flops / max(tm.n_launch, 1),
                 "algorithmic_bytes_per_launch": tm.bytes_min / max(tm.n_launch, 1),
                 # (a sampling timer -- the GEMM one, KernelTimer.sample -- has seen every `stride`-th launch only)
                 "time_share_of_timed_region": sec * max(int(getattr(tm, 'stride', 1)), 1) / max(elapsed, 1e-12)}
            r.update(extra)
            return r
        pmc_note = {"traffic_note": "no PMC summary for this workload (profiles/r04_svd_call_pmc.json is the chi=2048 Heisenberg call)"}
        pmc_file = os.path.join(ROOT, 'profiles', 'r06_svd_call_pmc.json')
        if not os.path.exists(pmc_file):
            pmc_file = os.path.join(ROOT, 'profiles', 'r05_svd_call_pmc.json')
        if args.config == 'heis2048' and chi == CONFIGS['heis2048'][1] and os.path.exists(pmc_file):
            with open(pmc_file) as f:
                pmc = json.load(f)
            pmc_note = {"traffic": pmc["bytes_per_call_corrected"],
                        # numerator and denominator of the SAME call (the sweep-average bytes are a different workload)
                        "traffic_call_algorithmic_bytes": pmc.get("algorithmic_bytes_same_call"),
                        "traffic_over_algorithmic": (pmc["bytes_per_call_corrected"] / pmc["algorithmic_bytes_same_call"])
                        if pmc.get("algorithmic_bytes_same_call") else None,
                        "traffic_note": "bytes per launch from separate rocprofv3 --pmc passes of the same call (FETCH_SIZE x 2 (gfx950) + WRITE_SIZE; "
                                        "%s: one COLD call on the saturated centre-bond theta), not collected in this "
                                        "run -- rocprofv3 counters cannot be read from inside the timed process; fabric-side counters incl. "
                                        "Infinity-Cache hits: the Gram matrices and accumulated transforms of the Gram-only rounds + the trailing updates of the pivoted QR, on-die"
                                        % ("profiles/" + os.path.basename(pmc_file))}
        roof_svd = roof(npc.svd_timer, "block SVD of one npc.svd call, all charge blocks together: cold = tpa_svd_batch (rank-revealing pivoted QR "
                                       "qrp_panel / qrp_update + one-sided Jacobi on 32-row blocks: Gram-only sweeps = Gram GEMM, svd_b32_round (solve + Gram / Qtot "
                                       "tile updates, one launch per round; complex: svd_b32_solve_c + svd_b32_gupdate_c), apply GEMM; + Q application), "
                                       "warm = failed or successful warm attempt (3 grouped GEMMs + the same Jacobi without the QR); + Loewdin clean-up GEMMs + "
                                       "basis store: ONE timer entry per npc.svd",
                        {**pmc_note,
                         "flop_model": "4 m^2 n + 8 m n^2 + 9 n^3 per block (m >= n), x4 for complex128 (SURVEY 8(d))"})
        roof_gemm = roof(npc.gemm_timer, "gemm_chain_kernel<f64 | c128> (grouped chained MFMA GEMM: tensordot / Lanczos matvec / env update)",
                         {"flop_model": "sum over GEMM links of c m k n, c = 2 real / 8 complex"})
        roof_eigh = None
        if npc.eigh_timer.n_launch:
            roof_eigh = roof(npc.eigh_timer, "tpa_eigh_batch (two-sided block Jacobi eigensolver of Hermitian blocks: _eig_based_svd / density-matrix mixer)",
                             {"flop_model": "9 n^3 per Hermitian n x n block, x4 for complex128"})
        n_upd = (L - 1) if is_tebd else 2 * (L - 2)
        unit = "s/step" if is_tebd else "s/sweep"
        what = "TEBD evolve_step time (s), TFI" if is_tebd else \
            "DMRG sweep time (s) + GS energy err, " + ("Heisenberg" if args.config == 'heis2048' else args.config)
        out = {"metric": "%s L=%d chi=%d" % (what, L, chi),
               "value": s_per_step, "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": 1e3 * s_per_step, "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
               "dtype": "c128" if is_tebd else "f64",
               "data": "synthetic (random right-canonical MPS at full chi)" if is_tebd else
                       "synthetic (state grown on-device from the Neel product state by an untimed chi ramp; no dataset/checkpoint)",
               "config": {"workload": "%s, L=%d, chi_max=%d; %s; 1 step = %d bond updates"
                          % (desc, L, chi, ("order-2 Suzuki-Trotter step, svd_min=1e-12" if is_tebd else
                                            "two-site DMRG sweep, Lanczos N=%d per bond, svd_min=1e-14, no mixer, combine=True "
                                            "interface (theta fused for the SVD)" % args.lanczos_N), n_upd),
                          "name": args.config,
                          "parallelism": "1 GPU" if world == 1 else
                          ("bonds of every half-step dealt over %d GPUs, one broadcast per new tensor" % world if (is_tebd and not args.qr) else
                           "%d replicas" % world if is_tebd else
                           "strong scaling of ONE chain: matvec row-sharded over %d GPUs (1 all-gather), SVD blocks dealt out (LPT + all-gather), rest "
                           "replicated; predicted Amdahl bound ~1.15x/1.25x/1.3x on 2/4/8 GPUs (DESIGN 5)" % world)},
               "prep_s": t_prep, "roofline": roof_svd, "roofline_gemm": roof_gemm, "energy_err": None}
        if roof_eigh is not None:
            out["roofline_eigh"] = roof_eigh
            if roof_eigh["time_share_of_timed_region"] > roof_svd["time_share_of_timed_region"]:      # --eig-svd: the eigensolver is the dominant family
                out["roofline"], out["roofline_svd"] = roof_eigh, roof_svd
        if is_tebd:
            out["tebd_route"] = ("QR-based truncation (decompose_theta_qr_based%s)" % (" + _eig_based_svd" if args.eig_svd else "")) if args.qr \
                else "block SVD of theta (svd_theta)"
            out["trunc_err_eps"] = float(eng.trunc_err.eps)
            out["norm"] = float(eng.norm)
            out["S_mid_entropy"] = float(-np.sum(np.asarray(eng.psi.get_SL(L // 2)) ** 2 * np.log(np.asarray(eng.psi.get_SL(L // 2)) ** 2 + 1e-300)))
            tref_file = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r03_cpu_reference_tebd.json')
            qr_ref = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r04_cpu_reference_tebd_qr.json')
            like_for_like = not args.qr
            eig_ref = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r06_cpu_reference_tebd_qr_eig.json')
            if args.qr and not args.eig_svd and os.path.exists(qr_ref):      # the reference's own QRBasedTEBDEngine, run offline (round 4)
                tref_file, like_for_like = qr_ref, True
            elif args.qr and args.eig_svd and os.path.exists(eig_ref):       # ... with use_eig_based_svd=True (truncation.py:473; round 6)
                tref_file, like_for_like = eig_ref, True
            n_done = args.warmup + args.steps
            if os.path.exists(tref_file) and world == 1:
                with open(tref_file) as f:
                    tref = json.load(f)
                if tref.get('L') == L and tref.get('chi') == chi:
                    out["cpu_baseline"] = {"value": tref['s_per_step_best'], "unit": "s/step", "cores": tref.get('cores'), "kind": "reference",
                                           "where": "offline: build container, %s host cores" % tref.get('cores'),
                                           "sample": "TeNPy's own " + tref.get('engine', 'TEBDEngine').split(' ')[0] + " (order 2, compiled _npc_helper) on the same synthetic state: best of "
                                                     "%d full steps (%s)" % (len(tref['steps']), os.path.basename(tref_file))}
                    rs = [r for r in tref['steps'] if r['step'] == n_done]
                    if rs:          # the reference has the state after exactly this many steps: compare what the line reports
                        r = rs[0]
                        Sd = np.sort(np.asarray(eng.psi.get_SL(L // 2)))[::-1]
                        out["tebd_parity"] = {
                            "after_steps": n_done, "reference": os.path.basename(tref_file),
                            "S_mid_entropy_abs_err": abs(out["S_mid_entropy"] - r['S_mid_entropy']),
                            "trunc_err_eps_rel_err": abs(out["trunc_err_eps"] - r['trunc_err_eps']) / max(abs(r['trunc_err_eps']), 1e-300),
                            "schmidt_top8_max_abs_err": float(np.max(np.abs(Sd[:8] - np.asarray(r['schmidt_top8'])))),
                            "chi_mid": [int(len(Sd)), int(r['chi_mid'])],
                            "like_for_like": like_for_like,
                            "note": "TeNPy's %s run offline on the same seeded state (scripts/cpu_reference_tebd.py)%s"
                                    % (tref.get('engine', 'TEBDEngine'), "" if like_for_like else "; the QR-based routes are different truncations "
                                       "and are expected to agree with the SVD-based reference only to the size of the truncation error")}
                    else:
                        out["tebd_parity"] = {"after_steps": n_done, "note": "the offline reference holds steps %s only: run with --warmup W --steps K, "
                                              "W + K among them, for the comparison" % [r['step'] for r in tref['steps']]}
        if not is_tebd:
            out["E"] = eng.sweep_stats['E'][-1]
            out["chi_reached"] = eng.sweep_stats['max_chi'][-1]
            out["E_sweeps"] = sweep_E
            out["untimed_sweeps"] = eng.ramp_log
            ref, ref_all = reference_entry(args.config, chi)
            if ref is not None and L == (80 if args.config == 'hubbard1024' else 100) and args.lanczos_N == 8:
                # the reference ran the same protocol; its sweep list starts with the same ramp
                ref_E = [e['E'] for e in ref['sweeps']]
                idx = min(len(sweep_E), len(ref_E)) - 1
                out["energy_err"] = abs(sweep_E[idx] - ref_E[idx]) / abs(ref_E[idx])
                out["energy_err_note"] = ("|E - E_ref| / |E_ref| after sweep %d of the common protocol (ramp + %d sweeps at chi=%d); "
                                          "E_ref = %.13f from TeNPy (compiled helper) run offline, profiles/r02_cpu_reference.json"
                                          % (idx + 1, idx + 1 - n_ramp, chi, ref_E[idx]))
            else:
                out["energy_err_note"] = "no offline TeNPy run of this configuration in profiles/r02_cpu_reference.json"
            if os.environ.get('TPA_BENCH_PHASES'):   # diagnostic run only: the phase timers synchronise the device
                out["phases_s"] = {k: round(v / max(args.steps, 1), 4) for k, v in eng.phase_time.items()}
            from tenpy_amd.linalg import _svd_warm
            n_svd = max(npc.svd_stats['calls'] - svd_calls0, 1)
            out["svd_stats"] = {"calls_timed": n_svd, "jacobi_sweeps_per_call": (npc.svd_stats['sweeps'] - svd_sweeps0) / n_svd,
                                "max_block": npc.svd_stats['max_block'], "abs_floor": npc.SVD_ABS_FLOOR,
                                "floor_on_min": bool(npc.SVD_FLOOR_ON_MIN), "jacobi_rounds": _dyn_rounds(),
                                "warm": {k: (v - warm0.get(k, 0)) for k, v in _svd_warm.stats.items() if not k.startswith('e_rel')},
                                "note": "warm = calls started from the singular vectors this bond produced on its previous visit "
                                        "(no pivoted QR; linalg/_svd_warm.py); the rest took the cold path; sweep counts include the "
                                        "low-rank residual decompositions of warm calls"}
            out["lanczos_stats"] = {k: _kb.stats[k] - lanczos0[k] for k in _kb.stats}
            out["lanczos_stats"]["note"] = ("timed sweeps only; n_ill_conditioned = results whose norm differed from 1 by more than 1e-5 before the "
                                            "final normalisation (the forced N_min = N_max = %d runs past convergence), n_degenerate = results that "
                                            "cancelled to zero and were replaced by the start vector" % args.lanczos_N)
            if world == 1:
                upd_t = eng.update_stats['time'][n0:]
                mid = [t for i, t in zip(eng.update_stats['i0'][n0:], upd_t) if abs(i - L // 2) <= 1]
                gpu_bond_s = float(np.mean(mid)) if mid else s_per_step / n_upd
                base = None
                if ref_all is not None and args.config in ('heis2048', 'xxz512', 'hubbard1024'):
                    envi = ref_all.get('environment', {})
                    if ref is not None:
                        base = {"value": ref['s_per_sweep_best'], "unit": unit, "cores": envi.get('cores'), "kind": "reference",
                                "where": "offline: build container, %s host cores" % envi.get('cores'),
                                "sample": "TeNPy %s with its compiled _npc_helper (OpenBLAS), set_level(3): best FULL sweep at "
                                          "chi=%d of the same protocol (%s)" % (envi.get('tenpy'), chi, os.path.basename(CPU_REF))}
                    elif chi == 2048 and 'bond2048' in ref_all:
                        b = ref_all['bond2048']
                        base = {"value": b['s_per_sweep_extrapolated'], "unit": unit, "cores": envi.get('cores'), "kind": "reference",
                                "where": "offline: build container, %s host cores" % envi.get('cores'),
                                "sample": "TeNPy %s, compiled helper: %.2f s per centre-bond update at chi=2048 (operands with the "
                                          "block structure of the real state) x 196 (%s)"
                                          % (envi.get('tenpy'), b['s_per_bond_best'], os.path.basename(CPU_REF))}
                if not args.no_cpu_baseline and args.config == 'heis2048' and chi == 2048:
                    live = reference_same_run()
                    if live is not None:
                        if base is not None:
                            live["offline_full_sweep"] = {k: base[k] for k in ("value", "unit", "cores", "where", "sample")}
                        base = live
                if not args.no_cpu_baseline:
                    try:
                        port, parity = parity_and_port(eng, args, gpu_bond_s)
                        out.update(parity)
                    except Exception as e:  # the baseline must never kill the bench line
                        port = {"value": None, "unit": unit, "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
                    if base is None:
                        base = port             # no offline TeNPy number for this configuration: the live oracle is the baseline
                    else:
                        base["port"] = port
                if base is not None:
                    out["cpu_baseline"] = base
        if world == 1 and not args.no_extras and args.config == 'heis2048' and not is_tebd and chi == CONFIGS['heis2048'][1]:
            try:
                extras(out, eng, args)
            except Exception as e:      # the extra legs must never kill the bench line
                out["extras_error"] = repr(e)
        if force_dist:
            out["config"]["parallelism"] = ("1 GPU, torch.distributed group of ONE rank over %s with the row-sharded operator forced: the all-gather of "
                                            "the N > 1 path runs as op kind 3 of tpa_lanczos_run (%d native sharded Lanczos runs)"
                                            % ("RCCL" if backend == 'nccl' else backend, out.get("lanczos_stats", {}).get("n_native_sharded", 0)))
        if emit:
            emit_lines(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return out


# ---- output: the driver parses the LAST stdout line and keeps only the tail of stdout, so the contract line must be short and
# ---- last (round 4's single 27 KB line was cut off: BENCH_r04.json "parsed": null).  Everything long goes out BEFORE it.
MAX_LINE = 4000


def _r(x, sig=6):
    """Floats to `sig` significant digits (JSON size), containers recursively; everything else unchanged."""
    if isinstance(x, float):
        return float('%.*g' % (sig, x)) if np.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def _roof_compact(r, name):
    if not isinstance(r, dict):
        return None
    keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_launch_ms", "algorithmic_flops_per_launch",
            "algorithmic_bytes_per_launch", "time_share_of_timed_region", "traffic_call_algorithmic_bytes", "traffic_over_algorithmic")
    c = {k: r[k] for k in keep if k in r}
    c["kernel"] = name
    return c


def compact(out):
    """The contract line: the keys the driver and the judge read, numbers only, no prose beyond short labels (< MAX_LINE bytes,
    asserted by tests/test_bench_contract.py on round 4's full line)."""
    c = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                             "vs_baseline", "dtype") if k in out}
    c["data"] = "synthetic"
    cfg = out.get("config", {})
    c["config"] = {"workload": (cfg.get("workload") or "")[:230], "name": cfg.get("name"), "parallelism": (cfg.get("parallelism") or "")[:200]}
    tebd = str(out.get("unit")) == "s/step"
    c["roofline"] = _roof_compact(out.get("roofline"), "tpa_svd_batch / tpa_svd_theta (+ clean-up): per npc.svd, all charge blocks")
    if "roofline_eigh" in out and out.get("roofline") is out.get("roofline_eigh"):
        c["roofline"]["kernel"] = "tpa_eigh_batch: Hermitian block eigensolver (_eig_based_svd)"
    c["roofline_gemm"] = _roof_compact(out.get("roofline_gemm"), "gemm_chain kernels: grouped chained MFMA GEMM")
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        c["cpu_baseline"] = {k: (cb[k][:200] if isinstance(cb[k], str) else cb[k]) for k in ("value", "unit", "cores", "kind", "sample") if k in cb}
        off = cb.get("offline_full_sweep")
        if isinstance(off, dict):
            c["cpu_baseline"]["offline_full_sweep"] = {"value": off.get("value"), "cores": off.get("cores")}
        port = cb.get("port")
        if isinstance(port, dict):
            c["cpu_baseline"]["port"] = {"value": port.get("value"), "cores": port.get("cores")}
    for k in ("energy_err", "E", "chi_reached", "sv_max_rel_err", "sv_max_rel_err_individual", "sv_kept", "sv_kept_rel_err_over_1e-10",
              "sv_kept_abs_err_over_32eps_smax", "mps_isometry_defect",
              "svd_isometry_defect", "matvec_max_rel_err", "E0_rel_err", "trunc_err_eps", "S_mid_entropy", "tebd_route", "prep_s"):
        if k in out:
            c[k] = out[k]
    hp = out.get("sv_vs_highprec")
    if isinstance(hp, dict) and "sv_max_rel_err_vs_highprec" in hp:      # (the per-decade table stays in the bench_detail line)
        c["sv_max_rel_err_vs_highprec"] = hp["sv_max_rel_err_vs_highprec"]
        c["lapack_max_rel_err_vs_highprec"] = hp["lapack_max_rel_err_vs_highprec"]
        c["sv_vs_highprec_worst_ratio_to_lapack"] = hp["worst_ratio_to_lapack"]
    tp = out.get("tebd_parity")
    if isinstance(tp, dict):
        c["tebd_parity"] = {k: tp[k] for k in ("after_steps", "S_mid_entropy_abs_err", "trunc_err_eps_rel_err", "schmidt_top8_max_abs_err",
                                               "like_for_like") if k in tp}
    ss = out.get("svd_stats")
    if isinstance(ss, dict):
        w = ss.get("warm", {})
        c["svd_stats"] = {"calls": ss.get("calls_timed"), "jacobi_sweeps_per_call": ss.get("jacobi_sweeps_per_call"), "max_block": ss.get("max_block"),
                          "warm_calls": w.get("warm_calls"), "sketch_calls": w.get("sketch_calls"), "cold_calls": w.get("cold_calls"),
                          "sk_residual": w.get("sk_residual"), "fb_stale": w.get("fb_stale"),
                          "sweeps_per_call": {k: round(w.get(k + "_sweeps", 0) / max(w.get(k + "_calls", 0), 1), 2) for k in ("warm", "sketch", "cold")},
                          "abs_floor": ss.get("abs_floor"), "floor_on_min": ss.get("floor_on_min"), "jacobi_rounds": ss.get("jacobi_rounds")}
    ls = out.get("lanczos_stats")
    if isinstance(ls, dict):
        c["lanczos_stats"] = {k: v for k, v in ls.items() if k != "note"}
    if "untimed_sweeps" in out:
        c["untimed_sweeps_s"] = [r_["s"] for r_ in out["untimed_sweeps"]]
    fs = out.get("first_sweeps_at_target_chi")
    if isinstance(fs, dict):
        c["first_sweeps_at_target_chi_s"] = fs.get("s")
    la = out.get("lanczos_adaptive")
    if isinstance(la, dict):
        c["lanczos_adaptive"] = {"s_per_sweep": la.get("s_per_sweep"), "matvecs_per_bond": la.get("matvecs_per_bond"), "E": la.get("E")}
    rv = out.get("roofline_vec")
    if isinstance(rv, dict) and "frac" in rv:
        c["roofline_vec"] = {"bound": "hbm", "achieved": rv.get("achieved"), "peak": rv.get("peak"), "unit": rv.get("unit"), "frac": rv.get("frac"),
                             "MB": rv.get("MB"), "kernels_frac": {k: v.get("frac") for k, v in rv.get("kernels", {}).items()}}
    oc = out.get("other_configs")
    if isinstance(oc, dict):
        c["other_configs"] = {}
        for name, o in oc.items():
            if "error" in o:
                c["other_configs"][name] = {"error": str(o["error"])[:80]}
                continue
            e = {"value": o.get("value")}
            if o.get("value_sweeps_3_4_at_target_chi") is not None:
                e["value_2p2"] = o["value_sweeps_3_4_at_target_chi"]
            for rk, ek in (("roofline", "frac"), ("roofline_gemm", "gemm_frac"), ("roofline_svd", "svd_frac")):
                if isinstance(o.get(rk), dict) and o[rk].get("frac"):
                    e[ek] = o[rk].get("frac")
            if isinstance(o.get("roofline"), dict):
                e["ms_per_call"] = o["roofline"].get("avg_launch_ms")
            for k in ("energy_err", "sv_max_rel_err"):
                if o.get(k) is not None:
                    e[k] = o[k]
            if isinstance(o.get("tebd_parity"), dict) and "S_mid_entropy_abs_err" in o["tebd_parity"]:
                e["entropy_abs_err"] = o["tebd_parity"]["S_mid_entropy_abs_err"]
            c["other_configs"][name] = e
    mf = out.get("module_form")
    if isinstance(mf, dict):
        c["module_form"] = {k: mf[k] for k in ("s_per_sweep_steady", "s_per_sweep_at_target_chi", "E", "vs_standalone", "skipped", "error") if k in mf}
        if isinstance(mf.get("svd_warm"), dict):      # (the full dictionary and the ramp sweeps: bench_detail line)
            c["module_form"]["svd_calls_warm_sketch_cold"] = [mf["svd_warm"].get(k) for k in ("warm_calls", "sketch_calls", "cold_calls")]
        if "error" in c["module_form"]:
            c["module_form"]["error"] = str(c["module_form"]["error"])[:120]
    fd = out.get("force_dist")
    if isinstance(fd, dict):
        c["force_dist"] = {k: (str(fd[k])[:100] if k == "error" else fd[k]) for k in ("s_per_sweep", "vs_value", "energy_err", "n_native_sharded", "error") if k in fd}
    for k in ("extras_s", "extras_error"):
        if k in out:
            c[k] = out[k] if k == "extras_s" else str(out[k])[:120]
    c["detail"] = "bench_detail lines above; copy under profiles/"
    c = _r(c)
    for k in ("E",):                        # energies keep every digit (the parity claim is 1e-10 relative)
        if k in out:
            c[k] = out[k]
    if isinstance(la, dict) and "lanczos_adaptive" in c:
        c["lanczos_adaptive"]["E"] = la.get("E")
    if isinstance(mf, dict) and "E" in mf and "module_form" in c:
        c["module_form"]["E"] = mf["E"]
    line = json.dumps(c, separators=(',', ':'))
    if len(line) > MAX_LINE:                 # never let optional parts cost the headline: drop them in this order
        for k in ("untimed_sweeps_s", "lanczos_stats", "roofline_vec", "force_dist", "svd_stats", "other_configs", "module_form", "cpu_baseline"):
            if k == "cpu_baseline":
                c[k] = {kk: vv for kk, vv in c.get(k, {}).items() if kk != "sample"}
            else:
                c.pop(k, None)
            line = json.dumps(c, separators=(',', ':'))
            if len(line) <= MAX_LINE:
                break
    return c, line


def emit_lines(out):
    """stdout: first the long material, one JSON object per line, each tagged ``bench_detail`` (the full headline with its notes, every
    extra leg); then -- LAST -- the compact contract line.  A copy of everything goes to gpurun_out/bench_full.json when that
    directory exists (builder runs; it is what lands under profiles/)."""
    big = ("other_configs", "module_form", "lanczos_adaptive", "roofline_vec", "first_sweeps_at_target_chi", "force_dist")
    head = {k: v for k, v in out.items() if k not in big}
    print(json.dumps({"bench_detail": "headline", **head}), flush=True)
    for k in big:
        if k not in out:
            continue
        if k == "other_configs" and isinstance(out[k], dict):
            for name, o in out[k].items():
                print(json.dumps({"bench_detail": "other_configs." + name, **o}), flush=True)
        else:
            print(json.dumps({"bench_detail": k, k: out[k]}), flush=True)
    try:
        d = os.path.join(ROOT, 'gpurun_out')
        if os.path.isdir(d):
            with open(os.path.join(d, 'bench_full.json'), 'w') as f:
                json.dump(out, f)
    except Exception:
        pass
    _, line = compact(out)
    print(line, flush=True)


def vector_roofline(n_elems, reps=20):
    """HBM roofline of the bandwidth-bound kernels of the path (K2-K4 Lanczos vector operations, K8-K10 copies) on
    vectors of the size of the timed wave function: HIP events around `reps` launches of each kernel on torch's current stream,
    algorithmic bytes as SURVEY 8(d) counts them (axpy 2R + 1W, scal 1R + 1W, dot 2R, norm 1R, fused Lanczos update 3R + 1W,
    packed copy 1R + 1W)."""
    import torch
    from tenpy_amd.linalg import _device as dev
    L = dev.lib()
    st = dev.stream()
    n = int(n_elems)
    x, y, z = (torch.randn(n, dtype=torch.float64, device='cuda') for _ in range(3))
    scr = torch.zeros(4096, dtype=torch.float64, device='cuda')
    res = torch.zeros(8, dtype=torch.float64, device='cuda')
    jobs = np.zeros((1, 4 + 3 * 6), dtype=np.int64)
    jobs[0, 2], jobs[0, 4], jobs[0, 5] = 2, n // 1024, 1024            # one 2-D job: rows x 1024, unit strides
    jobs[0, 4 + 6], jobs[0, 5 + 6], jobs[0, 4 + 12], jobs[0, 5 + 12] = 1024, 1, 1024, 1
    jd = dev.to_device(jobs)
    kernels = {
        "axpy": (3, lambda: L.tpa_axpy(0, n, 0.5, 0.0, x.data_ptr(), y.data_ptr(), st)),
        "scal": (2, lambda: L.tpa_scal(0, n, 1.0000001, 0.0, y.data_ptr(), st)),
        "dot": (2, lambda: L.tpa_dot(0, n, x.data_ptr(), y.data_ptr(), 0, res.data_ptr(), scr.data_ptr(), st)),
        "nrm2sq": (1, lambda: L.tpa_nrm2sq(0, n, x.data_ptr(), res.data_ptr(), scr.data_ptr(), st)),
        "lanczos_update": (4, lambda: L.tpa_lanczos_update(0, n, z.data_ptr(), -0.3, 0.0, x.data_ptr(), -0.2, 0.0, y.data_ptr(),
                                                           res.data_ptr(), scr.data_ptr(), st)),
        "copy_batch": (2, lambda: L.tpa_copy_batch(0, jd.data_ptr(), 1, (n // 1024) * 1024, x.data_ptr(), z.data_ptr(), st)),
    }
    per, tot_b, tot_ms = {}, 0., 0.
    for name, (passes, fn) in kernels.items():
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        nbytes = passes * 8. * (n if name != "copy_batch" else (n // 1024) * 1024)
        per[name] = {"us": round(1e3 * ms, 2), "GBs": round(nbytes / ms / 1e6, 1), "frac": round(nbytes / ms / 1e6 / PEAK_HBM_GBS, 4)}
        tot_b += nbytes
        tot_ms += ms
    ach = tot_b / tot_ms / 1e6
    return {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS, "traffic": None,
            "vector_elements": n, "kernels": per,
            "note": "Lanczos vector kernels and the packed copy on float64 vectors of the size of the timed wave function (%d elements = "
                    "%.1f MB): HIP events around %d launches each, algorithmic bytes per SURVEY 8(d); at this size the vectors stay in the "
                    "256 MB Infinity Cache, so a fraction above 1 of the HBM peak is cache bandwidth, not an error" % (n, 8e-6 * n, reps)}


def extras(out, eng, args):
    """Legs after the headline (one GPU, heis2048 only; untimed for `value`), so that the driver's line witnesses what earlier rounds
    only had in builder-run files under profiles/ (VERDICT r3 task 2): the first sweeps at the target chi, one sweep with the
    reference's adaptive Lanczos rule, the HBM roofline of the vector kernels, the other BASELINE configurations, and TeNPy's own
    TwoSiteDMRGEngine on the device."""
    import subprocess
    import torch
    from tenpy_amd.linalg import krylov_based as kb
    t_all = time.time()
    ramp = out.get("untimed_sweeps", [])
    out["first_sweeps_at_target_chi"] = {
        "s": [r["s"] for r in ramp if r.get("chi_max") == args.chi][:3],
        "note": "the first sweeps after chi_max reached %d (every bond still growing, few warm SVD starts): the non-steady-state regime; "
                "`value` is the steady-state average" % args.chi}
    # ---- one sweep with the reference's own Lanczos rule (dmrg.py:302: N_min = 2, N_max = 20, P_tol = 1e-14)
    saved = eng.lanczos_params
    l0 = dict(kb.stats)
    eng.lanczos_params = {'N_min': 2, 'N_max': 20, 'P_tol': 1.e-14}
    t_ad = []
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        eng.sweep()
        torch.cuda.synchronize()
        t_ad.append(time.time() - t0)
    eng.lanczos_params = saved
    st = {k: kb.stats[k] - l0[k] for k in kb.stats}
    out["lanczos_adaptive"] = {"s_per_sweep": t_ad[-1], "sweeps_s": t_ad, "E": float(eng.sweep_stats['E'][-1]), "lanczos_stats": st,
                               "matvecs_per_bond": st.get('n_matvec', 0) / max(4 * (eng.psi.L - 2), 1),
                               "note": "two further sweeps on the same state with the reference's adaptive stopping rule (N_min=2, N_max=20, "
                                       "P_tol=1e-14, dmrg.py:302) instead of the forced N=%d of the timed sweeps; s_per_sweep = the second one "
                                       "(in the first the change of protocol invalidates the warm-start bases of the block SVD)" % args.lanczos_N}
    try:
        out["roofline_vec"] = vector_roofline(sum(int(b.size) for b in eng.psi.get_theta(eng.psi.L // 2 - 1, n=2)._data))
    except Exception as e:
        out["roofline_vec"] = {"error": repr(e)}
    # ---- the other BASELINE configurations, in this process (no second import of torch)
    keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "roofline_gemm", "energy_err",
            "energy_err_note", "sv_max_rel_err", "sv_max_rel_err_individual", "svd_isometry_defect", "matvec_max_rel_err", "E0_rel_err",
            "tebd_parity", "tebd_route", "cpu_baseline", "prep_s", "svd_stats", "roofline_eigh", "roofline_svd")
    others = {}
    for name, argv in (("xxz512", ["--config", "xxz512"]), ("hubbard1024", ["--config", "hubbard1024"]),
                       ("tebd1024", ["--config", "tebd1024"]), ("tebd1024_qr", ["--config", "tebd1024", "--qr"]),
                       ("tebd1024_qr_eig", ["--config", "tebd1024", "--qr", "--eig-svd"])):
        t0 = time.time()
        try:
            is_t = name.startswith("tebd")
            # DMRG configurations: the headline's protocol (5 warm-up sweeps at the target chi, then the timed ones); the figure of the
            # 2 + 2 sweep legs of rounds 4 - 6 (sweeps 3 and 4 at the target chi: still a third cold SVD calls) is kept beside it
            r = run(argv + (["--steps", "2", "--warmup", "1"] if is_t else ["--steps", "3", "--warmup", "5"]) + ["--no-extras", "--cpu-sample-bonds", "1"], emit=False)
            o = {k: r[k] for k in keep if k in r}
            if not is_t:
                tgt = [e["s"] for e in r.get("untimed_sweeps", []) if str(e.get("kind", "")).startswith("warm-up at the target chi")]
                if len(tgt) >= 4:
                    o["value_sweeps_3_4_at_target_chi"] = float(np.mean(tgt[2:4]))
            if "svd_stats" in o:
                o["svd_stats"] = {k: v for k, v in o["svd_stats"].items() if k in ("calls_timed", "jacobi_sweeps_per_call", "max_block")}
            if "cpu_baseline" in o and isinstance(o["cpu_baseline"], dict):
                o["cpu_baseline"] = {k: v for k, v in o["cpu_baseline"].items() if k != "port"}
            o["leg_s"] = round(time.time() - t0, 1)
            others[name] = o
        except Exception as e:
            others[name] = {"error": repr(e)}
        torch.cuda.synchronize()
    out["other_configs"] = others
    # ---- TeNPy's own engine (the reference archive, unmodified) on the device: a second process (the import hook replaces modules)
    try:
        from oracle import build_ref
        if build_ref.reference_root() is None:
            out["module_form"] = {"skipped": "no reference tree / archive on this machine"}
        else:
            t0 = time.time()
            pr = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'module_form_bench.py'), '--chi', str(args.chi), '--L',
                                 str(args.L), '--sweeps', '5'], capture_output=True, text=True, timeout=400)
            line = [ln for ln in pr.stdout.splitlines() if ln.startswith('{')]
            if pr.returncode != 0 or not line:
                out["module_form"] = {"error": (pr.stderr or pr.stdout)[-400:]}
            else:
                m = json.loads(line[-1])
                tgt = [r for r in m["sweeps"] if r["kind"].startswith("target")]
                ts = [r["s"] for r in tgt]
                steady = float(np.mean(ts[-3:])) if len(ts) >= 5 else None
                out["module_form"] = {
                    "what": m["what"], "s_per_sweep_at_target_chi": ts, "E": tgt[-1]["E"] if tgt else None,
                    # VERDICT r4 task 7: TeNPy's own engine in the regime the headline is quoted in (>= 5 sweeps at the target chi, mean of the last 3)
                    "s_per_sweep_steady": steady, "vs_standalone": (steady / out["value"]) if steady else None,
                    "ramp_sweeps_s": [r["s"] for r in m["sweeps"] if not r["kind"].startswith("target")],
                    "two_site_h": m.get("two_site_h"),
                    "svd_warm": {k: v for k, v in (m.get("svd_warm") or {}).items() if k in ("warm_calls", "sketch_calls", "cold_calls", "fallbacks", "fb_stale", "fb_nomatch", "sk_residual")},
                    "leg_s": round(time.time() - t0, 1),
                    "note": "tenpy.algorithms.dmrg.TwoSiteDMRGEngine of the reference archive, unmodified, tenpy_amd.install.install(fused=True): "
                            "the bench protocol (Neel state, mixer on during the chi ramp, then 5 sweeps at the target chi with Lanczos N=8; "
                            "s_per_sweep_steady = mean of the last 3, vs_standalone = that / `value`) in a "
                            "second process of this run; energies comparable with `untimed_sweeps` / profiles/r02_cpu_reference.json"}
    except Exception as e:
        out["module_form"] = {"error": repr(e)}
    # ---- the N > 1 code path at world size 1 (VERDICT r5 task 6): row-sharded operator, all-gather from the collective callback of the
    #      native Lanczos run, distributed SVD -- over RCCL with one rank, in a second process; what it costs against `value`
    try:
        t0 = time.time()
        pr = subprocess.run([sys.executable, os.path.abspath(__file__), '--force-dist', '--chi', str(args.chi), '--L', str(args.L),
                             '--steps', '2', '--warmup', '3', '--no-extras', '--no-cpu-baseline'], capture_output=True, text=True, timeout=400)
        line = [ln for ln in pr.stdout.splitlines() if ln.startswith('{"metric"')]
        if pr.returncode != 0 or not line:
            out["force_dist"] = {"error": (pr.stderr or pr.stdout)[-300:]}
        else:
            m = json.loads(line[-1])
            out["force_dist"] = {"s_per_sweep": m["value"], "vs_value": m["value"] / out["value"], "energy_err": m.get("energy_err"),
                                 "n_native_sharded": (m.get("lanczos_stats") or {}).get("n_native_sharded"),
                                 "parallelism": (m.get("config") or {}).get("parallelism"), "leg_s": round(time.time() - t0, 1),
                                 "note": "the sharded (N > 1) path with ONE rank over RCCL: 3 + 2 sweeps at the target chi in a second process; "
                                         "no multi-GPU node in the pool -- the scaling curve stays unmeasured"}
    except Exception as e:
        out["force_dist"] = {"error": repr(e)}
    out["extras_s"] = round(time.time() - t_all, 1)


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        return so.getsockname()[1]


def _dyn_rounds():
    """Rounds the activity-driven schedule of the Gram-only sweeps launched against the round-robin count (whole process)."""
    try:
        import ctypes
        from tenpy_amd import _lib
        o = (ctypes.c_int64 * 3)()
        _lib.load().tpa_svd_dyn_stats(o, 0)
        return {"launched": int(o[0]), "round_robin": int(o[1]), "sweeps": int(o[2])}
    except Exception:      # the emulated device has no such counters
        return None


def main():
    """``python bench.py --gpus N`` with N > 1 and no rendezvous in the environment launches its own N ranks (one per GPU) through
    ``torch.distributed.run`` on 127.0.0.1 and passes the command line on; under the driver's launcher (WORLD_SIZE set) it is a rank."""
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        import subprocess
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
               '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus != world:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher's world size is what runs and what n_gpus reports\n" % (args.gpus, world))
    run()


if __name__ == '__main__':
    main()
