"""Debug helper: the matrices of tests/test_kernels_gpu.py::test_svd_refinement_steps through tpa_svd_batch with the refinement end game,
per block: orthonormality defects of U and VH (sigma > 1e-6 sigma_max) and the refinement counters."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from tenpy_amd import _lib
from test_kernels_gpu import _svd_call
lib = _lib.load()
pre = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g = torch.Generator(device="cpu").manual_seed(5 + pre)
dt = torch.float64
mats = []
for (m, n, r) in [(300, 300, 160), (200, 333, 90), (420, 390, 390), (70, 40, 1), (128, 257, 128), (5, 90, 5)]:
    u, _ = torch.linalg.qr(torch.randn(m, r, dtype=dt, generator=g))
    v, _ = torch.linalg.qr(torch.randn(n, r, dtype=dt, generator=g))
    sv = torch.logspace(0, -9, r, dtype=torch.float64)
    if r > 8:
        sv[r // 3] = sv[r // 3 + 1] = sv[r // 3 + 2]
        sv[5] = sv[4] * (1 - 1e-9)
    mats.append((u * sv.to(dt)) @ v.conj().T)
for code, name in ((2097152 | ((pre + 1) << 16), "refine pre=%d" % pre), (0, "default"), (2097152 | ((pre + 1) << 16) | 8388608, "refine, 2 launches per round"),
                   (2097152 | ((pre + 1) << 16) | 1048576, "refine, rounds on the data")):
    out = (ctypes.c_int64 * 8)()
    lib.tpa_svd_refine_stats(out, 1)
    lib.tpa_svd_set_algorithm(code)
    res, rc, sw = _svd_call(torch, lib, mats)
    lib.tpa_svd_set_algorithm(0)
    lib.tpa_svd_refine_stats(out, 1)
    print(name, "rc", rc, "sweeps+steps", sw, "stats", list(out))
    for x, (u, s, vh) in zip(mats, res):
        nz = s > 1e-6 * s[0]
        k = int(nz.sum())
        eu = (u[:, nz].T @ u[:, nz] - torch.eye(k, dtype=dt)).abs().max().item() if k else 0
        ev = (vh[nz] @ vh[nz].T - torch.eye(k, dtype=dt)).abs().max().item() if k else 0
        ref = torch.linalg.svdvals(x)
        print("   block %s k=%d  |UtU-1| %.1e  |VVt-1| %.1e  dS %.1e recon %.1e" % (tuple(x.shape), k, eu, ev, (s - ref).abs().max().item() / ref[0].item(),
              ((u * s) @ vh - x).abs().max().item()))
