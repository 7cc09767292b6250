"""Dev harness for the batched block SVD: loads the charge blocks of a dumped theta (npz, default the saturated chi=2048
Heisenberg theta), runs tpa_svd_batch, prints time / sweeps / accuracy against LAPACK (host, checker only).
The default file is not tracked (25 MB); regenerate it on a GPU box with
``TPA_DUMP_THETA=scripts/data/theta_chi2048_sat.npz python bench.py --steps 1``."""
import ctypes
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tenpy_amd import _lib

lib = _lib.load()
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), 'data', 'theta_chi2048_sat.npz')
d = np.load(path)
blocks = [np.ascontiguousarray(d[k]) for k in d.files]
CHECK = int(os.environ.get('CHECK', 1))
REPS = int(os.environ.get('REPS', 5))
jobs, a_off, s_off, u_off, v_off = [], 0, 0, 0, 0
for b in blocks:
    m, n = b.shape
    k = min(m, n)
    jobs.append([a_off, m, n, u_off, s_off, v_off, 0, 0])
    a_off += m * n
    u_off += m * k
    v_off += k * n
    s_off += k
A = torch.from_numpy(np.concatenate([b.reshape(-1) for b in blocks])).cuda()
jh = np.array(jobs, np.int64)
refS = [np.linalg.svd(b, compute_uv=False) for b in blocks] if CHECK else None
for alg in [int(x) for x in os.environ.get('ALGS', '0').split(',')]:
    lib.tpa_svd_set_algorithm(alg)
    U = torch.zeros(u_off, dtype=torch.float64, device='cuda')
    VH = torch.zeros(v_off, dtype=torch.float64, device='cuda')
    S = torch.zeros(s_off, dtype=torch.float64, device='cuda')
    wb = lib.tpa_svd_worksize(0, jh.ctypes.data, len(jobs))
    work = torch.empty(wb, dtype=torch.uint8, device='cuda')
    sw = ctypes.c_int()
    st = torch.cuda.current_stream().cuda_stream
    rho = float(os.environ.get('RHO', 1e-6))
    ts = []
    for rep in range(REPS + 1):
        torch.cuda.synchronize()
        t0 = time.time()
        rc = lib.tpa_svd_batch(0, jh.ctypes.data, len(jobs), A.data_ptr(), U.data_ptr(), S.data_ptr(), VH.data_ptr(),
                               work.data_ptr(), wb, 80, rho, ctypes.byref(sw), st)
        torch.cuda.synchronize()
        ts.append(time.time() - t0)
    msg = "alg=%d rc=%d sweeps=%d time=%.2f ms (min %.2f)" % (alg, rc, sw.value, 1e3 * np.mean(ts[1:]), 1e3 * min(ts[1:]))
    if CHECK:
        Sh, Uh, Vh = S.cpu().numpy(), U.cpu().numpy(), VH.cpu().numpy()
        es, rec, orthu, orthv = 0., 0., 0., 0.
        for b, j, sr in zip(blocks, jobs, refS):
            m, n = b.shape
            k = min(m, n)
            s = Sh[j[4]:j[4] + k]
            u = Uh[j[3]:j[3] + m * k].reshape(m, k)
            v = Vh[j[5]:j[5] + k * n].reshape(k, n)
            es = max(es, np.abs(np.sort(s)[::-1] - sr).max() / sr.max())
            rec = max(rec, np.abs((u * s) @ v - b).max() / np.abs(b).max())
            keep = s > 1e-14 * s.max()
            orthu = max(orthu, np.abs(u[:, keep].T @ u[:, keep] - np.eye(keep.sum())).max())
            orthv = max(orthv, np.abs(v[keep] @ v[keep].T - np.eye(keep.sum())).max())
        msg += "  |dS|/Smax=%.2e recon=%.2e |UtU-1|=%.2e |VVt-1|=%.2e (kept: sigma>1e-14 max)" % (es, rec, orthu, orthv)
    print(msg, flush=True)
lib.tpa_svd_set_algorithm(0)
