"""tpa_eigh_batch: the direct two-sided iteration (TPA_EIGH_DIRECT=1, default) against the shift + one-sided route (=0) on batches of
Hermitian blocks: time, sweeps, residual |A V - V w|, orthogonality, eigenvalues against LAPACK.
usage: eigh_direct_bench.py [real|complex] [n] [blocks] [flat|graded]"""
import os
import sys
import time
import ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tenpy_amd.linalg import _device as dev

kind = sys.argv[1] if len(sys.argv) > 1 else 'complex'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 8
spec = sys.argv[4] if len(sys.argv) > 4 else 'flat'
rng = np.random.default_rng(1)
dt = np.complex128 if kind == 'complex' else np.float64
mats = []
for b in range(nb):
    x = rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if kind == 'complex' else 0)
    if spec == 'graded14':        # full rank, 14 decades (tests/test_kernels_gpu.py::test_eigh_batch_mixer_blocks)
        q, _ = np.linalg.qr(x)
        h = (q * np.logspace(0, -14, n)) @ q.conj().T
        h = 0.5 * (h + h.conj().T)
    elif spec == 'graded':        # rank-deficient graded PSD: the mixer's density matrix
        r = n // 2
        y = x[:, :r] * np.logspace(0, -8, r)
        h = y @ y.conj().T
    else:                          # Xi^dagger Xi of a flat-spectrum bond matrix
        h = x.conj().T @ x / n
    mats.append(np.ascontiguousarray(h.astype(dt)))
flat = np.concatenate([m.reshape(-1) for m in mats])
L = dev.lib()
code = dev.code(dt)
jobs = np.zeros((nb, 8), dtype=np.int64)
jobs[:, 0] = np.arange(nb) * n * n
jobs[:, 1] = n
jobs[:, 2] = np.arange(nb) * n
jobs[:, 3] = np.arange(nb) * n * n
a_dev = dev.to_device(flat)
wb = L.tpa_eigh_worksize(code, jobs.ctypes.data, nb)
work = torch.empty(int(wb), dtype=torch.uint8, device='cuda')
wref = [np.linalg.eigvalsh(m) for m in mats[:2]]
L.tpa_svd_set_algorithm(int(os.environ.get('ALG', '0')))
for direct in (1, 0, 1):
    L.tpa_eigh_set_direct(direct)
    W = dev.empty(nb * n, np.float64)
    V = dev.empty(nb * n * n, dt)
    sw = ctypes.c_int()
    t = 1e9
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.time()
        dev.check(L.tpa_eigh_batch(code, jobs.ctypes.data, nb, a_dev.data_ptr(), W.data_ptr(), V.data_ptr(), work.data_ptr(), int(wb), 60, 0.0,
                                   ctypes.byref(sw), dev.stream()), "eigh")
        torch.cuda.synchronize()
        t = min(t, time.time() - t0)       # (min of 5: one-off ~70 ms stalls of the runtime in the first second of a process)
    w = dev.to_host(W).reshape(nb, n)
    v = dev.to_host(V).reshape(nb, n, n)
    res = orth = werr = 0.
    for b in range(min(nb, 2)):
        sc = np.linalg.norm(mats[b])
        res = max(res, np.abs(mats[b] @ v[b] - v[b] * w[b][None, :]).max() / sc)
        orth = max(orth, np.abs(v[b].conj().T @ v[b] - np.eye(n)).max())
        werr = max(werr, np.abs(w[b] - wref[b]).max() / sc)
    fl = 9. * n ** 3 * nb * (4 if kind == 'complex' else 1)
    print("%s n=%d blocks=%d %s direct=%d: %.2f ms, %d sweeps, %.2f TFLOP/s (9 n^3 model)  |AV-Vw|/|A| %.1e  |V^H V - 1| %.1e  |w - w_lapack|/|A| %.1e"
          % (kind, n, nb, spec, direct, t * 1e3, sw.value, fl / t / 1e12, res, orth, werr), flush=True)
L.tpa_eigh_set_direct(1)
