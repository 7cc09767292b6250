"""Micro-benchmark of the grouped chained GEMM: (a) one dense fp64 GEMM, (b) the two tensordots of a
TwoSiteH matvec on a synthetic Sz-conserving block structure of bond dimension chi."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tenpy_amd.linalg import np_conserved as npc
from tenpy_amd.linalg.charges import ChargeInfo, LegCharge, LegPipe


def dense(n, reps=int(os.environ.get('REPS', '40'))):
    ch = ChargeInfo()
    leg = LegCharge.from_trivial(n, ch)
    fill = os.environ.get('FILL', 'rand')     # 'ones': constant data -> low toggle power (DVFS check)
    gen = (lambda: np.random.rand(n, n)) if fill == 'rand' else (lambda: np.full((n, n), 1.0))
    a = npc.Array.from_ndarray(gen(), [leg, leg.conj()])
    b = npc.Array.from_ndarray(gen(), [leg, leg.conj()])
    plan, a, b = npc.plan_tensordot(a, b, axes=1)
    out = None
    for _ in range(2):
        out = plan.apply(a, b)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        out = plan.apply(a, b)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / reps
    print("dense[%s] %d^3: %.3f ms  %.2f TFLOP/s  (tiles %d)" % (fill, n, dt * 1e3, 2 * n**3 / dt / 1e12, plan.n_tiles), flush=True)


def sectors(chi, var=8.0):
    q = np.arange(-12, 13, 2) + 1   # odd 2Sz sectors
    w = np.exp(-q**2 / (2 * var))
    n = np.maximum((w / w.sum() * chi).astype(int), 0)
    n[len(n) // 2] += chi - n.sum()
    keep = n > 0
    return q[keep], n[keep]


def matvec(chi, reps=int(os.environ.get('REPS', '40'))):
    ch = ChargeInfo([1])
    q, n = sectors(chi)
    vL = LegCharge.from_qind(ch, np.concatenate([[0], np.cumsum(n)]), q.reshape(-1, 1), qconj=+1)
    q2, n2 = q + 1, n     # other bond: shifted by one spin
    vR = LegCharge.from_qind(ch, np.concatenate([[0], np.cumsum(n2)]), q2.reshape(-1, 1) - 1, qconj=-1)
    p = LegCharge.from_qflat(ch, [[-1], [1]])
    w = LegCharge.from_qflat(ch, [[0], [2], [-2], [0], [0]], qconj=-1)
    pipeL = LegPipe([vL, p], qconj=+1)
    pipeR = LegPipe([p, vR], qconj=-1)
    rnd = lambda sh: np.random.standard_normal(sh)
    LHeff = npc.Array.from_func(rnd, [pipeL, w, pipeL.conj()], labels=['(vR*.p0)', 'wR', '(vR.p0*)'])
    RHeff = npc.Array.from_func(rnd, [w.conj(), pipeR.conj(), pipeR], labels=['wL', '(p1*.vL)', '(p1.vL*)'])
    theta = npc.Array.from_func(rnd, [pipeL, pipeR], labels=['(vL.p0)', '(p1.vR)'])
    p1, _, _ = npc.plan_tensordot(LHeff, theta, axes=['(vR.p0*)', '(vL.p0)'])
    tmp = p1.apply(LHeff, theta)
    p2, _, _ = npc.plan_tensordot(tmp, RHeff, axes=(['wR', '(p1.vR)'], ['wL', '(p1*.vL)']))
    for _ in range(2):
        tmp = p1.apply(LHeff, theta)
        res = p2.apply(tmp, RHeff)
    torch.cuda.synchronize()
    ts = []
    for pl, args in ((p1, (LHeff, theta)), (p2, (tmp, RHeff))):
        t0 = time.time()
        for _ in range(reps):
            pl.apply(*args)
        torch.cuda.synchronize()
        ts.append((time.time() - t0) / reps)
    fl = p1.flops + p2.flops
    for nm, pl, t in (('step1', p1, ts[0]), ('step2', p2, ts[1])):
        bm, bn = npc._gemm_tile(pl.dtype, pl.cfg)
        tk, lk = pl.tasks_host, pl.links_host
        kpad = np.array([np.sum(-(-lk[f:f + c, 2] // 16) * 16) for f, c in zip(tk[:, 4], tk[:, 5])])
        mm, nn = tk[:, 1], tk[:, 2]
        pads = {}
        for g in (bm, 64, 16):       # rows/cols padded to the tile, to 64 and to 16
            gm, gn = min(g, bm), min(g, bn)
            pads[g] = float(np.sum(2.0 * (-(-mm // gm) * gm) * (-(-nn // gn) * gn) * kpad))
        sk = getattr(pl, 'sk', None)
        print("   %s cfg %d tile %dx%d tiles %d%s: useful %.2f GF; padded to tile %.3fx, to 64 %.3fx, to 16 %.3fx; rate on tile-padded work %.1f TF/s" % (
            nm, pl.cfg, bm, bn, pl.n_tiles, (' split-K %d tiles' % sk.n_tiles) if sk is not None else '', pl.flops / 1e9,
            pads[bm] / pl.flops, pads[64] / pl.flops, pads[16] / pl.flops, pads[bm] / t / 1e12), flush=True)
    print("matvec chi=%d: sectors %s  step1 %.3f ms (%.1f TF/s, %d gemms, %d tiles)  step2 %.3f ms (%.1f TF/s, %d gemms, %d tiles)  "
          "total %.3f ms = %.2f TFLOP/s, %.1f GB min traffic" % (
              chi, n.tolist(), ts[0] * 1e3, p1.flops / ts[0] / 1e12, p1.n_gemm, p1.n_tiles, ts[1] * 1e3, p2.flops / ts[1] / 1e12,
              p2.n_gemm, p2.n_tiles, sum(ts) * 1e3, fl / sum(ts) / 1e12, (p1.bytes_min + p2.bytes_min) / 1e9), flush=True)


if __name__ == '__main__':
    from tenpy_amd import _lib
    if os.environ.get('GEMM_VARIANT'):
        _lib.load().tpa_gemm_set_variant(int(os.environ['GEMM_VARIANT']))      # (TPA_GEMM_VARIANT in the environment does the same for any program)
    if os.environ.get('XCD_ORDER'):
        npc.XCD_TILE_ORDER = bool(int(os.environ['XCD_ORDER']))
    if os.environ.get('GEMM_CFG'):
        npc.FORCE_GEMM_CFG = int(os.environ['GEMM_CFG'])
    for n in [int(x) for x in os.environ.get('DENSE', '1024,2048,4096').split(',') if x]:
        dense(n)
    for chi in [int(x) for x in os.environ.get('CHIS', '512,1024,2048').split(',') if x]:
        matvec(chi)
