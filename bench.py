#!/usr/bin/env python
"""bench.py -- two-site DMRG sweep time on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--chi CHI] [--L L]

Workload (config["workload"]): spin-1/2 Heisenberg chain (XXZ, Jxx=Jz=1, hz=0, Sz conserved), L=100,
two-site DMRG at bond dimension chi (default 2048, BASELINE.json's headline configuration; fits one
GPU), Lanczos with N_min=N_max=8 (fixed work per bond, as tests/benchmark/dmrg_infinite.py:36 of the
reference does), svd_min=1e-14 (the reference's default, so that the bond dimension really saturates at chi), no mixer.  A "step" is ONE FULL SWEEP = 2(L-2) = 196 two-site bond
updates (effective-H build, 8-step Lanczos, block SVD + truncation, environment update).

The MPS is synthetic in the sense of the contract: there is no checkpoint to load, so the state is grown
on the device from the Neel product state by an (untimed) chi ramp of single sweeps
chi = 64, 128, ..., chi/2, followed by W warm-up sweeps at the target chi; then exactly K sweeps are
timed between barrier + torch.cuda.synchronize() on both sides.

Multi-GPU (--gpus N, launched by torch.distributed.run, one rank per GPU over RCCL): the Lanczos matvec is
sharded over the ranks by rows of theta' (tenpy_amd/algorithms/sharded.py: row panels of both GEMM steps +
one all-gather per matvec), the charge blocks of every SVD are distributed (one all-gather); environment update and
the Lanczos vector kernels are replicated.  Total work is fixed -> "scaling": "strong"; the value is the max over ranks of the time per sweep.

One JSON line on rank 0; `roofline` is for the grouped MFMA GEMM (tensordot / Lanczos matvec kernel),
`cpu_baseline` is the numpy oracle (oracle/npc_oracle.py) timed on the host cores on a bounded sample.
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--chi', type=int, default=int(os.environ.get('TPA_BENCH_CHI', 2048)))
    ap.add_argument('--L', type=int, default=100)
    ap.add_argument('--lanczos-N', type=int, default=8)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample-bonds', type=int, default=2)
    return ap.parse_args()


def oracle_tensor(arr):
    """Device Array -> oracle OTensor (host copies of the blocks)."""
    from oracle import npc_oracle as orc
    legs = [orc.OLeg(l.slices, l.charges, l.qconj, arr.chinfo.mod) for l in arr.legs]
    return orc.OTensor(legs, arr.qtotal, arr._qdata, arr._data)


def cpu_baseline(eng, args, gpu_bond_s):
    """Time the numpy oracle on `cpu_sample_bonds` centre-bond updates of the SAME state: 8 effective-H
    matvecs (two block-sparse tensordots each) + the block SVD of theta.  Returns the cpu_baseline dict."""
    from oracle import npc_oracle as orc
    from tenpy_amd.algorithms.mps_common import TwoSiteH
    L = eng.psi.L
    n_b = max(1, args.cpu_sample_bonds)
    bonds = [L // 2 - 1 + i for i in range(n_b)]
    t_cpu, errs = 0., []
    for i0 in bonds:
        eff = TwoSiteH(eng.env, i0, factored=False)       # the fused form LHeff . theta . RHeff that the oracle restates
        theta = eff.combine_theta(eng.psi.get_theta(i0, n=2))
        fac = TwoSiteH(eng.env, i0)                       # what the timed sweeps ran (factored when W has scalar blocks)
        want = fac.prepare_svd(fac.matvec(fac.combine_theta(eng.psi.get_theta(i0, n=2))))
        LH, RH, th = oracle_tensor(eff.LHeff), oracle_tensor(eff.RHeff), oracle_tensor(theta)
        t0 = time.time()
        v = th
        for _ in range(args.lanczos_N):
            v = orc.matvec_two_site(LH, RH, v)
            nv = orc.norm(v)
            v = orc.scale(v, 1. / nv)
        blocked, _ = orc.combine_legs(th, [[0], [1]], [th.legs[0].qconj, th.legs[1].qconj]) \
            if not _legs_blocked(th) else (th, None)
        orc.svd(blocked)
        t_cpu += time.time() - t0
        first = orc.matvec_two_site(LH, RH, th)
        ref = first.to_dense()
        errs.append(float(np.max(np.abs(want.to_ndarray() - ref)) / max(np.max(np.abs(ref)), 1e-300)))
    per_bond = t_cpu / n_b
    n_bonds = 2 * (L - 2)
    return {"value": per_bond * n_bonds, "unit": "s/sweep", "cores": os.cpu_count(), "kind": "port",
            "sample": "%d centre bond updates (%d Lanczos matvecs + block SVD each) with the numpy oracle on the "
                      "same state, %.2f s CPU per bond, extrapolated x%d bonds; GPU same bonds %.4f s per bond; "
                      "max |matvec_gpu - matvec_oracle| / max|.| = %.2e" % (n_b, args.lanczos_N, per_bond, n_bonds,
                                                                            gpu_bond_s, max(errs)),
            "matvec_max_rel_err": max(errs)}


def _legs_blocked(t):
    return all(len({tuple(c) for c in l.charges.tolist()}) == len(l.charges) for l in t.legs)


def main():
    args = parse()
    import torch
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    backend = os.environ.get('TPA_BENCH_BACKEND', 'nccl')     # 'gloo': CPU dry run of the N>1 control flow (tests/test_bench_contract.py)
    if world > 1:
        import torch.distributed as dist
        if backend == 'nccl':
            torch.cuda.set_device(local)
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend=backend)
    else:
        dist = None
        torch.cuda.set_device(0)

    from tenpy_amd.algorithms.dmrg import TwoSiteDMRGEngine
    from tenpy_amd.linalg import np_conserved as npc
    from tenpy_amd.models.spin_chains import spin_half_leg, xxz_chain_mpo
    from tenpy_amd.networks.mps import MPS

    L, chi = args.L, args.chi
    H = xxz_chain_mpo(L, 1., 1., 0.)
    _, p = spin_half_leg('Sz')
    psi = MPS.from_product_state([p] * L, [1, 0] * (L // 2))
    eng = TwoSiteDMRGEngine(psi, H, {'trunc_params': {'chi_max': min(64, chi), 'svd_min': 1.e-14},
                                     'lanczos_params': {'N_min': 2, 'N_max': 20}, 'shard_matvec': world > 1,
                                     'profile': bool(os.environ.get('TPA_BENCH_PHASES'))})
    # ---- untimed: grow the state (two quick sweeps at small chi, then double chi per sweep)
    t_prep = time.time()
    c = min(64, chi)
    eng.sweep()
    eng.sweep()
    while c < chi:
        c = min(2 * c, chi)
        eng.trunc_params['chi_max'] = c
        if c == chi:
            break
        eng.sweep()
    eng.lanczos_params = {'N_min': args.lanczos_N, 'N_max': args.lanczos_N}
    for _ in range(args.warmup):
        eng.sweep()
    torch.cuda.synchronize()
    t_prep = time.time() - t_prep

    eng.phase_time = {k: 0. for k in eng.phase_time}
    # ---- timed region: exactly K sweeps
    npc.gemm_timer.reset()
    npc.gemm_timer.enabled = True
    n0 = len(eng.update_stats['E_total'])
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        eng.sweep()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.time() - t0
    npc.gemm_timer.enabled = False
    gemm_ms = npc.gemm_timer.collect()
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    s_per_sweep = elapsed / max(args.steps, 1)
    E = eng.sweep_stats['E'][-1]
    chi_reached = eng.sweep_stats['max_chi'][-1]

    if rank == 0:
        gt = npc.gemm_timer
        tflops = (gt.flops / (gemm_ms * 1e-3)) / 1e12 if gemm_ms > 0 else 0.
        per_launch_ms = gemm_ms / max(gt.n_launch, 1)
        traffic, traffic_note = None, None
        try:     # L2-miss traffic of the matvec GEMM measured with rocprofv3 PMC passes (committed profile, not live)
            with open(os.path.join(ROOT, 'profiles', 'r01_gemm_pmc_matvec_factored_chi2048.json')) as f:
                pm = json.load(f)
            if chi == 2048:
                traffic, traffic_note = pm["traffic_bytes_per_launch"], pm["how"]
        except Exception:
            pass
        roof = {"bound": "mfma", "achieved": tflops, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": tflops / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic, "traffic_note": traffic_note,
                "kernel": "gemm_chain_kernel<f64, 64x64 | 128x128> (grouped chained MFMA GEMM: tensordot / Lanczos matvec / env update)",
                "matvec_form": "factored: LP.theta (GEMM) -> W0 W1 blockwise (lincomb) -> .RP (GEMM); d=2 times fewer flops than LHeff.theta.RHeff",
                "launches": gt.n_launch, "avg_launch_ms": per_launch_ms, "algorithmic_flops_per_launch": gt.flops / max(gt.n_launch, 1),
                "algorithmic_bytes_per_launch": gt.bytes_min / max(gt.n_launch, 1),
                "time_share_of_sweep": (gemm_ms * 1e-3) / max(elapsed, 1e-12)}
        upd_t = eng.update_stats['time'][n0:]
        mid = [t for i, t in zip(eng.update_stats['i0'][n0:], upd_t) if abs(i - L // 2) <= 1]
        gpu_bond_s = float(np.mean(mid)) if mid else s_per_sweep / (2 * (L - 2))
        out = {"metric": "DMRG sweep time (s), Heisenberg L=%d chi=%d" % (L, chi), "value": s_per_sweep, "unit": "s/sweep",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * s_per_sweep,
               "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
               "data": "synthetic (state grown on-device from the Neel product state by an untimed chi ramp; no dataset/checkpoint)",
               "config": {"workload": "two-site DMRG sweep, spin-1/2 Heisenberg chain (XXZ Jxx=Jz=1, Sz conserved), L=%d, "
                                      "chi_max=%d (reached %d), Lanczos N=%d per bond, svd_min=1e-14, no mixer, combine=True interface (theta fused for the SVD; matvec applied in factored form); 1 step = 1 sweep = %d bond updates"
                                      % (L, chi, chi_reached, args.lanczos_N, 2 * (L - 2)),
                          "parallelism": "1 GPU" if world == 1 else "matvec row-sharded over %d GPUs (all-gather per matvec), SVD charge blocks distributed (LPT + all-gather), env update replicated" % world},
               "E": E, "chi_reached": chi_reached, "prep_s": t_prep, "roofline": roof}
        if os.environ.get('TPA_BENCH_PHASES'):   # diagnostic run only: the phase timers synchronise the device
            out["phases_s"] = {k: round(v / max(args.steps, 1), 4) for k, v in eng.phase_time.items()}
            out["svd_stats"] = dict(npc.svd_stats)
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N=1 only (the other ranks would idle at the barrier)
            try:
                out["cpu_baseline"] = cpu_baseline(eng, args, gpu_bond_s)
            except Exception as e:  # the baseline must never kill the bench line
                out["cpu_baseline"] = {"value": None, "unit": "s/sweep", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
