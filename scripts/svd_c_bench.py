"""Complex block SVD (tpa_svd_batch, c128) on the shape of a TEBD bond of config 5 (two 1024 x 1024 parity blocks with a flat spectrum):
Gram-only sweeps on 32-row blocks (default, csrc/tpa_svd_b32c.inc) against the 8-row-block rounds on the data (bit 20)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from tenpy_amd import _lib
from test_kernels_gpu import _svd_call
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
g = torch.Generator(device="cpu").manual_seed(3)
mats = []
for b in range(2):
    x = torch.randn(n, n, dtype=torch.complex128, generator=g)
    u, _ = torch.linalg.qr(x)
    v, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.complex128, generator=g))
    sv = torch.logspace(0, -4, n, dtype=torch.float64)
    mats.append((u * sv.to(torch.complex128)) @ v.conj().T)
for code, name in ((0, "Gram-only sweeps, 32-row blocks"), (1048576, "8-row-block rounds on the data")):
    lib.tpa_svd_set_algorithm(code)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res, rc, sw = _svd_call(torch, lib, mats)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    lib.tpa_svd_set_algorithm(0)
    x, (u, s, vh) = mats[0], res[0]
    ref = torch.linalg.svdvals(x)
    print("%-34s rc %d sweeps %d  %.1f ms per call (incl. host transfers of the harness)  dS %.1e recon %.1e |UhU-1| %.1e |VVh-1| %.1e" % (
        name, rc, sw, 1e3 * dt, (s - ref).abs().max().item(), ((u * s.to(u.dtype)) @ vh - x).abs().max().item(),
        (u.conj().T @ u - torch.eye(n, dtype=u.dtype)).abs().max().item(), (vh @ vh.conj().T - torch.eye(n, dtype=u.dtype)).abs().max().item()))
