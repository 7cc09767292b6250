"""Time / check the batched device SVD on the sector structure of a chi=2048 Heisenberg theta."""
import ctypes
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tenpy_amd import _lib

lib = _lib.load()
sizes = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [1086, 872, 872, 450, 450, 148, 148, 31, 31, 4, 4]
decay = float(sys.argv[2]) if len(sys.argv) > 2 else 12.
RANKFRAC = float(os.environ.get('RANKFRAC', 1.))
CPLX = bool(int(os.environ.get('CPLX', 0)))
DT = torch.complex128 if CPLX else torch.float64
g = torch.Generator().manual_seed(1)
mats, specs = [], []
for n in sizes:
    u, _ = torch.linalg.qr(torch.randn(n, n, dtype=DT, generator=g))
    v, _ = torch.linalg.qr(torch.randn(n, n, dtype=DT, generator=g))
    s = torch.logspace(0, -decay, n, dtype=torch.float64)
    if RANKFRAC < 1.:    # numerically rank deficient like a DMRG theta block: exact zeros beyond the rank
        s[max(1, int(RANKFRAC * n)):] = 0.
    mats.append((u * s.to(DT)) @ v.conj().T)
    specs.append(s)
jobs, a_off, s_off = [], 0, 0
for n in sizes:
    jobs.append([a_off, n, n, a_off, s_off, a_off, 0, 0])
    a_off += n * n
    s_off += n
A = torch.cat([m.reshape(-1) for m in mats]).cuda()
jh = np.array(jobs, np.int64)
import itertools
rhos = [float(x) for x in os.environ.get('RHOS', '0,1e-4,1e-2,1').split(',')]
for alg, rho in itertools.product([int(x) for x in os.environ.get('ALGS', '32,16,1').split(',')], rhos):
    lib.tpa_svd_set_algorithm(alg)
    U = torch.zeros(a_off, dtype=DT, device='cuda')
    VH = torch.zeros(a_off, dtype=DT, device='cuda')
    S = torch.zeros(s_off, dtype=torch.float64, device='cuda')
    wb = lib.tpa_svd_worksize(int(CPLX), jh.ctypes.data, len(jobs))
    work = torch.empty(wb, dtype=torch.uint8, device='cuda')
    sw = ctypes.c_int()
    st = torch.cuda.current_stream().cuda_stream
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        rc = lib.tpa_svd_batch(int(CPLX), jh.ctypes.data, len(jobs), A.data_ptr(), U.data_ptr(), S.data_ptr(), VH.data_ptr(),
                               work.data_ptr(), wb, 80, rho, ctypes.byref(sw), st)
        torch.cuda.synchronize()
        dt = time.time() - t0
    errs, recs, orth = [], [], []
    for b, n in enumerate(sizes):
        o = jobs[b][0]
        u = U[o:o + n * n].reshape(n, n)
        vh = VH[o:o + n * n].reshape(n, n)
        s = S[jobs[b][4]:jobs[b][4] + n]
        errs.append(float(((s.cpu() - specs[b]).abs() / specs[b].clamp_min(1e-300)).max()))
        recs.append(float(((u * s.to(DT)) @ vh - mats[b].cuda()).abs().max()))
        orth.append(float((u.conj().T @ u - torch.eye(n, dtype=DT, device='cuda')).abs().max()))
    abs_err = max(float((S[jobs[b][4]:jobs[b][4] + n].cpu() - specs[b]).abs().max()) for b, n in enumerate(sizes))
    print("alg=%s rho=%g rc=%d sweeps=%d time=%.1f ms  abs err S=%.2e  recon=%.2e  |U^TU-1|=%.2e" % (
        (('block/ls%d' % ((alg >> 4) & 15)) + ('' if alg & 512 else '+qrp')) if alg % 2 == 0 else 'pairwise', rho, rc, sw.value, dt * 1e3, abs_err, max(recs), max(orth)), flush=True)
lib.tpa_svd_set_algorithm(0)
