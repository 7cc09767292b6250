"""Where do the Jacobi sweeps of a WARM-started block SVD go?  (VERDICT r5 task 2b.)  Replays the device's default iteration
(tests/jacobi_emulation.py::jacobi_b32: Gram-only sweeps on 32-row blocks, activity-driven rounds, rule with the absolute floor on the
smaller row) on matrices W = Bq X^H that the MI355X run handed to `tpa_svd_batch` (TPA_SVD_DUMP_W; the largest charge block of a call)
and prints, at the START of every sweep, by band of decades of sigma_i / sigma_max of the two rows of a pair: how many pairs still fail
the stopping rule, how many of them are "big" (their rotation changes the Gram matrix beyond the predicted-convergence bound, i.e. the
sweep after this one is still needed), and the largest cosine.
    python scripts/warm_sweep_histogram.py W00.npy [W01.npy ...] [--rho 1e-2]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import jacobi_emulation as je

BANDS = [(0, 1), (2, 4), (5, 8), (9, 12), (13, 99)]


def table(W, tol, floor2, on_min):
    nrm2 = np.einsum('ij,ij->i', W, W)
    S = W @ W.T
    smax = np.sqrt(nrm2.max())
    with np.errstate(all='ignore'):
        dec = np.where(nrm2 > 0, np.floor(-np.log10(np.sqrt(nrm2) / smax + 1e-300)), 99).astype(int)
    band = np.zeros(len(W), dtype=int)
    for k, (lo, hi) in enumerate(BANDS):
        band[(dec >= lo) & (dec <= hi)] = k
    iu = np.triu_indices(len(W), 1)
    a, b, g = nrm2[iu[0]], nrm2[iu[1]], S[iu]
    nd = je.needs32(a, b, g * g, tol, floor2, on_min)
    bg = nd & je.big32(a, b, g * g, floor2, on_min)
    with np.errstate(all='ignore'):
        cos = np.abs(g) / np.sqrt(a * b)
    cos = np.where(np.isfinite(cos), cos, 0.)
    lo_, hi_ = np.minimum(band[iu[0]], band[iu[1]]), np.maximum(band[iu[0]], band[iu[1]])
    out = ["     rows per band (decades 0-1 | 2-4 | 5-8 | 9-12 | 13+): %s" % [int(np.sum(band == k)) for k in range(len(BANDS))]]
    for x in range(len(BANDS)):
        cells = []
        for y in range(x, len(BANDS)):
            sel = (lo_ == x) & (hi_ == y)
            if not sel.any():
                continue
            cells.append("%d x %d: %7d fail %7d big  max|cos| %.1e" % (x, y, int(nd[sel].sum()), int(bg[sel].sum()), cos[sel].max()))
        out.append("     " + "  |  ".join(cells))
    return "\n".join(out), int(nd.sum()), int(bg.sum())


def run(path, rho, on_min=True):
    W = np.load(path)
    R, L = W.shape
    fro2 = float((W * W).sum())
    tol, floor2 = je.EPS * np.sqrt(L), rho * rho * fro2
    print("== %s: %d x %d, rho %g, floor on the %s row" % (os.path.basename(path), R, L, rho, 'smaller' if on_min else 'larger'))
    sweep = 0
    rounds_total = 0
    while True:
        txt, n_need, n_big = table(W, tol, floor2, on_min)
        print("   start of sweep %d: %d pairs fail the rule, %d big" % (sweep, n_need, n_big))
        print(txt)
        if n_need == 0:
            break
        # one sweep of the default iteration = jacobi_b32 limited to one sweep
        sw, rounds, W, _ = je.jacobi_b32(W, rho, on_min, max_sweeps=1)
        rounds_total += rounds
        print("   sweep %d ran %d rounds (round-robin would run %d)" % (sweep, rounds, max((R + 31) // 32 // 2 * 2 + ((R + 31) // 32) % 2 * 2 - 1, 1)))
        sweep += 1
        if n_big == 0 or sweep > 30:
            break
    s = np.sort(np.linalg.norm(W, axis=1))[::-1]
    ref = np.linalg.svd(np.load(path), compute_uv=False)
    print("   -> %d sweeps, %d rounds; sigma vs LAPACK max abs err / sigma_max %.1e" % (sweep, rounds_total, np.abs(s - ref[:len(s)]).max() / ref[0]))


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    rho = float(sys.argv[sys.argv.index('--rho') + 1]) if '--rho' in sys.argv else 1e-2
    args = [a for a in args if a != str(rho) and a != sys.argv[sys.argv.index('--rho') + 1]] if '--rho' in sys.argv else args
    for p in args:
        run(p, rho)
