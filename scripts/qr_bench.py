"""tpa_qr_batch: blocked compact-WY QR vs the one-workgroup Householder kernel on QR-TEBD sized blocks (d chi x 1.1 chi)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tenpy_amd import _lib

lib = _lib.load()
cplx = bool(int(os.environ.get('CPLX', 0)))
dt = torch.complex128 if cplx else torch.float64
shapes = [(int(a), int(b)) for a, b in (x.split('x') for x in (sys.argv[1] if len(sys.argv) > 1 else '2048x1126,2048x1126').split(','))]
g = torch.Generator().manual_seed(3)
mats = [torch.randn(m, n, dtype=dt, generator=g) for m, n in shapes]
jobs, a_off, q_off, r_off = [], 0, 0, 0
for m, n in shapes:
    k = min(m, n)
    jobs.append([a_off, m, n, q_off, r_off, 0, 0, 0])
    a_off, q_off, r_off = a_off + m * n, q_off + m * k, r_off + k * n
A = torch.cat([x.reshape(-1) for x in mats]).cuda()
Q = torch.zeros(q_off, dtype=dt).cuda()
R = torch.zeros(r_off, dtype=dt).cuda()
jh = np.array(jobs, np.int64)
st = torch.cuda.current_stream().cuda_stream
for mode, name in ((0, 'WY, 1 launch/panel'), (2, 'WY, 2 launches/panel'), (1, 'one workgroup')):
    lib.tpa_qr_set_algorithm(mode)
    if mode == 1 and max(m for m, _ in shapes) * (16 if cplx else 8) > 150 * 1024:
        continue                 # (the one-workgroup kernel keeps a column in LDS)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.time()
        rc = lib.tpa_qr_batch(int(cplx), jh.ctypes.data, len(jobs), A.data_ptr(), Q.data_ptr(), R.data_ptr(), st)
        torch.cuda.synchronize()
        dt_s = time.time() - t0
    m, n = shapes[0]
    k = min(m, n)
    q = Q[:m * k].reshape(m, k)
    r = R[:k * n].reshape(k, n)
    print("%-22s %s %s rc=%d  %.2f ms   |QR-A|=%.1e  |Q^HQ-1|=%.1e" % (name, 'c128' if cplx else 'f64', shapes, rc, dt_s * 1e3,
          float((q @ r - mats[0].cuda()).abs().max()), float((q.conj().T @ q - torch.eye(k, dtype=dt, device='cuda')).abs().max())), flush=True)
lib.tpa_qr_set_algorithm(0)
