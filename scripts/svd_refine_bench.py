"""Dev harness for the refinement end game of the block SVD (csrc/tpa_svd_refine.inc): the charge blocks of a dumped theta
(default: the saturated chi=2048 Heisenberg theta) through tpa_svd_batch with the refinement off / on after `pre` cyclic sweeps,
cold (pivoted QR) and in the shape of a warm-started call (rows pre-rotated by the exact singular vectors of a slightly perturbed
copy, no pivoted QR); prints time, steps, Newton-Schulz steps and accuracy against LAPACK (host, checker only)."""
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
REPS = int(os.environ.get('REPS', 5))
RHO = float(os.environ.get('RHO', 1e-6))


def stats(reset=True):
    out = (ctypes.c_int64 * 8)()
    lib.tpa_svd_refine_stats(out, int(reset))
    return list(out)


def run(blocks, alg, label, refS):
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
    lib.tpa_svd_set_algorithm(alg)
    U = torch.zeros(u_off, dtype=torch.float64, device='cuda')
    VH = torch.zeros(v_off, dtype=torch.float64, device='cuda')
    S = torch.zeros(s_off, dtype=torch.float64, device='cuda')
    wb = lib.tpa_svd_worksize(0, jh.ctypes.data, len(jobs))
    work = torch.empty(wb, dtype=torch.uint8, device='cuda')
    sw = ctypes.c_int()
    st = torch.cuda.current_stream().cuda_stream
    ts = []
    for rep in range(REPS + 1):
        stats()
        torch.cuda.synchronize()
        t0 = time.time()
        rc = lib.tpa_svd_batch(0, jh.ctypes.data, len(jobs), A.data_ptr(), U.data_ptr(), S.data_ptr(), VH.data_ptr(),
                               work.data_ptr(), wb, 80, RHO, ctypes.byref(sw), st)
        torch.cuda.synchronize()
        ts.append(time.time() - t0)
    stt = stats()
    lib.tpa_svd_set_algorithm(0)
    msg = "%-34s alg=%7d rc=%d sweeps+steps=%2d time %.2f ms (min %.2f) [refine calls %d steps %d NS %d pre-sweeps %d extra %d | plain calls %d sweeps %d | failed %d]" % (
        label, alg, rc, sw.value, 1e3 * np.mean(ts[1:]), 1e3 * min(ts[1:]), *stt)
    msg += " reps " + " ".join("%.1f" % (1e3 * t) for t in ts)
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
        keep = s > 1e-6 * s.max()
        orthu = max(orthu, np.abs(u[:, keep].T @ u[:, keep] - np.eye(keep.sum())).max())
        orthv = max(orthv, np.abs(v[keep] @ v[keep].T - np.eye(keep.sum())).max())
    print(msg + "  |dS|/Smax=%.1e recon=%.1e |UtU-1|=%.1e |VVt-1|=%.1e (sigma > 1e-6 max)" % (es, rec, orthu, orthv), flush=True)


refS = [np.linalg.svd(b, compute_uv=False) for b in blocks]
print("blocks:", [b.shape for b in blocks])
REF = 2097152          # bit 21: refinement steps on;  bit 20 (1048576): Gram-only sweeps off
if os.environ.get('ONLY_ALG'):       # one variant only (for a rocprofv3 kernel trace)
    run(blocks, int(os.environ['ONLY_ALG']), "alg %s" % os.environ['ONLY_ALG'], refS)
    sys.exit(0)
run(blocks, 1048576 | 4194304, "cold: round-3 rounds (data), round-3 solve", refS)
run(blocks, 4194304, "cold: Gram-only sweeps, round-3 solve", refS)
run(blocks, 1048576, "cold: round-3 rounds (data)", refS)
run(blocks, 8388608, "cold: Gram-only, 2 launches per round", refS)
run(blocks, 0, "cold: Gram-only, fused rounds", refS)
run(blocks, 8192, "cold: Gram-only, no look-ahead", refS)
for pre in (3, 2):
    run(blocks, REF | ((pre + 1) << 16), "cold: Gram-only %d sweeps + refinement" % pre, refS)
# the shape of a warm-started call: W = Bq X^H with Bq = right singular vectors of a copy perturbed at the 1e-9 level,
# numerical rank only -> short, wide blocks whose rows are nearly orthogonal
rng = np.random.RandomState(0)
warm = []
for b in blocks:
    u, s, vh = np.linalg.svd(b + 1e-9 * np.abs(b).max() * rng.standard_normal(b.shape), full_matrices=False)
    k = max(1, int(np.sum(s > 1e-15 * np.sqrt(np.sum(s * s)))))
    warm.append(np.ascontiguousarray(vh[:k] @ b.T))
refW = [np.linalg.svd(b, compute_uv=False) for b in warm]
print("warm-shaped blocks:", [b.shape for b in warm])
run(warm, 512 | 1048576, "warm-shaped: round-3 rounds", refW)
run(warm, 512 | 8388608, "warm-shaped: Gram-only, 2 launches", refW)
run(warm, 512, "warm-shaped: Gram-only, fused rounds", refW)
