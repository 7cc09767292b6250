"""tpa_svd_batch on dumped theta blocks (np.savez of the blocks; see svd_overhead.py DUMP_THETA)."""
import ctypes
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tenpy_amd import _lib

lib = _lib.load()
z = np.load(sys.argv[1])
mats = [torch.from_numpy(z[k]) for k in z.files]
if os.environ.get('SVD_ALG'):
    lib.tpa_svd_set_algorithm(int(os.environ['SVD_ALG']))
jobs, a_off, u_off, s_off, v_off = [], 0, 0, 0, 0
for x in mats:
    m, n = x.shape
    k = min(m, n)
    jobs.append([a_off, m, n, u_off, s_off, v_off, 0, 0])
    a_off, u_off, s_off, v_off = a_off + m * n, u_off + m * k, s_off + k, v_off + k * n
A = torch.cat([x.reshape(-1) for x in mats]).cuda()
U = torch.zeros(u_off, dtype=torch.float64).cuda()
S = torch.zeros(s_off, dtype=torch.float64).cuda()
VH = torch.zeros(v_off, dtype=torch.float64).cuda()
jh = np.array(jobs, np.int64)
wb = lib.tpa_svd_worksize(0, jh.ctypes.data, len(jobs))
work = torch.empty(wb, dtype=torch.uint8).cuda()
sw = ctypes.c_int()
st = torch.cuda.current_stream().cuda_stream
reps = int(os.environ.get('REPS', 3))
for rep in range(reps):
    torch.cuda.synchronize()
    t0 = time.time()
    rc = lib.tpa_svd_batch(0, jh.ctypes.data, len(jobs), A.data_ptr(), U.data_ptr(), S.data_ptr(), VH.data_ptr(),
                           work.data_ptr(), wb, 80, 1e-6, ctypes.byref(sw), st)
    torch.cuda.synchronize()
    dt = time.time() - t0
print("blocks", [tuple(x.shape) for x in mats][:8], "n =", len(mats))
print("rc=%d sweeps=%d time=%.2f ms" % (rc, sw.value, dt * 1e3), flush=True)
