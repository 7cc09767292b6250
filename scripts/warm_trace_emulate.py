"""Dev harness (CPU): numpy emulation of the Gram-only block-Jacobi sweeps on 32-row blocks (csrc/tpa_svd_b32.inc: one exact Gram
matrix per sweep, 17 rounds on it -- first round all 2016 local pairs, then cross pairs only --, one product with the accumulated
transform), same stopping rule / predicted-convergence rule as the kernels (svd_needs_rotation / svd_big_rotation), with a TRACE:
per sweep, by decade of sigma_i / sigma_max of the two rows of a pair, how many pairs still need a rotation and the largest cosine.
Input: matrices captured by scripts/warm_trace_capture.py (CPU) or dumped on the GPU (TPA_SVD_DUMP_W).
Usage: python scripts/warm_trace_emulate.py file.npz [rho] [max_matrices]"""
import sys
import numpy as np

EPS = 2.220446049250313e-16
BB, TB = 32, 64


def pair_of(R, pair, rnd):
    NB = (R + BB - 1) // BB
    NBp = (NB + 1) // 2 * 2
    mod = NBp - 1
    r = rnd % mod if mod > 0 else 0
    if pair == 0:
        bi, bj = NBp - 1, r
    else:
        bi, bj = (r + pair) % mod, (r - pair + mod) % mod
    if bi > bj:
        bi, bj = bj, bi
    return bi, bj, NB, NBp


def local_pairs(rr, full_local):
    a = np.arange(BB)
    if not full_local:
        return a, BB + ((a + rr) & (BB - 1))
    x = np.where(a == 0, TB - 1, (rr + a) % (TB - 1))
    y = np.where(a == 0, rr, (rr - a + (TB - 1)) % (TB - 1))
    return np.minimum(x, y), np.maximum(x, y)


def needs(a, b, g2, tol, floor2):
    mn, mx0 = np.minimum(a, b), np.maximum(a, b)
    ok = (a > 0) & (b > 0) & ~(mn < 1e-60 * mx0)
    return ok & (g2 > tol * tol * mn * np.maximum(mx0, floor2))


def big(a, b, g2, floor2):
    mn, mx0 = np.minimum(a, b), np.maximum(a, b)
    above = mx0 >= floor2
    return np.where(above, g2 > 1e-14 * mn * mx0, g2 > 1e-14 * mn * np.sqrt(mx0 * floor2))


def solve_pair(Sm, full_local, tol, floor2):
    """Cyclic two-sided Jacobi pass on the 64 x 64 Gram block; returns (Q, flag, flag_big) -- Q None if nothing needs a rotation."""
    iu = np.triu_indices(TB, 1) if full_local else np.nonzero(np.add.outer(np.arange(TB) < BB, np.zeros(TB, bool)) & (np.arange(TB) >= BB)[None, :])
    d = np.diag(Sm)
    a, b, g = d[iu[0]], d[iu[1]], Sm[iu]
    nd = needs(a, b, g * g, tol, floor2)
    if not nd.any():
        return None, False, False
    fb = bool((nd & big(a, b, g * g, floor2)).any())
    Q = np.eye(TB)
    Sm = Sm.copy()
    for rr in range(TB - 1 if full_local else BB):
        p, q = local_pairs(rr, full_local)
        al, be, ga = Sm[p, p], Sm[q, q], Sm[p, q]
        nr = needs(al, be, ga * ga, tol, floor2)
        if not nr.any():
            continue
        with np.errstate(all='ignore'):
            zeta = (be - al) / (2.0 * ga)
            t = np.copysign(1.0, zeta) / (np.abs(zeta) + np.sqrt(zeta * zeta + 1.0))
            c = 1.0 / np.sqrt(t * t + 1.0)
            s = c * t
        c = np.where(nr, c, 1.0)
        s = np.where(nr, s, 0.0)
        J = np.eye(TB)
        J[p, p], J[q, q], J[p, q], J[q, p] = c, c, -s, s
        Sm = J @ Sm @ J.T
        Q = J @ Q
    return Q, True, fb


def decade_table(W, tol, floor2, label):
    nrm2 = np.einsum('ij,ij->i', W, W)
    S = W @ W.T
    smax = np.sqrt(nrm2.max())
    with np.errstate(all='ignore'):
        dec = np.where(nrm2 > 0, np.floor(-np.log10(np.sqrt(nrm2) / smax + 1e-300)), 99).astype(int)
    dec = np.minimum(dec, 16)
    iu = np.triu_indices(len(W), 1)
    a, b, g = nrm2[iu[0]], nrm2[iu[1]], S[iu]
    nd = needs(a, b, g * g, tol, floor2)
    bg = nd & big(a, b, g * g, floor2)
    with np.errstate(all='ignore'):
        cos = np.abs(g) / np.sqrt(a * b)
    cos = np.where(np.isfinite(cos), cos, 0.)
    # bands of decades: 0-1, 2-4, 5-8, 9-12, 13+
    bands = [(0, 1), (2, 4), (5, 8), (9, 12), (13, 16)]
    def band(dv):
        out = np.zeros_like(dv)
        for k, (lo, hi) in enumerate(bands):
            out[(dv >= lo) & (dv <= hi)] = k
        return out
    bi, bj = band(dec[iu[0]]), band(dec[iu[1]])
    lo_, hi_ = np.minimum(bi, bj), np.maximum(bi, bj)
    rows = ["   %s rows per band %s" % (label, [int(np.sum(band(dec) == k)) for k in range(len(bands))])]
    for x in range(len(bands)):
        cells = []
        for y in range(x, len(bands)):
            sel = (lo_ == x) & (hi_ == y)
            cells.append("%d-%d:%6d need %6d big maxcos %.1e" % (x, y, int(nd[sel].sum()), int(bg[sel].sum()), cos[sel].max() if sel.any() else 0.))
        rows.append("     " + " | ".join(cells))
    return "\n".join(rows), int(nd.sum()), int(bg.sum())


def jacobi_trace(W0, rho=1e-2, max_sweeps=30, verbose=True, name=''):
    W = W0.copy()
    R, L = W.shape
    fro2 = float((W * W).sum())
    tol, floor2 = EPS * np.sqrt(L), rho * rho * fro2
    NB = (R + BB - 1) // BB
    NBp = (NB + 1) // 2 * 2
    rounds = max(NBp - 1, 1)
    active_hist = []
    for sweep in range(max_sweeps):
        if verbose:
            txt, n_need, n_big = decade_table(W, tol, floor2, 'sweep %d start:' % sweep)
            print(txt)
        S = W @ W.T
        Qtot = np.eye(R)
        cnt = nbig = 0
        act_blocks = np.zeros(NBp, int)
        for r in range(rounds):
            full_local = 1 if r == 0 else 0
            for pair in range(NBp // 2):
                bi, bj, _, _ = pair_of(R, pair, r)
                idx = np.concatenate([np.arange(bi * BB, bi * BB + BB), np.arange(bj * BB, bj * BB + BB)])
                ok = idx < R
                if bi >= NB:
                    ok[:BB] = False
                if bj >= NB:
                    ok[BB:] = False
                gi = np.where(ok, idx, 0)
                Sm = S[np.ix_(gi, gi)] * np.outer(ok, ok)
                Q, flag, fb = solve_pair(Sm, full_local, tol, floor2)
                if not flag:
                    continue
                cnt += 1
                nbig += int(fb)
                act_blocks[bi] += 1
                act_blocks[bj] += 1
                rows = gi[ok]
                Qs = Q[np.ix_(ok, ok)]
                S[rows, :] = Qs @ S[rows, :]
                S[:, rows] = S[:, rows] @ Qs.T
                Qtot[rows, :] = Qs @ Qtot[rows, :]
        W = Qtot @ W
        active_hist.append(act_blocks[:NB].copy())
        if verbose:
            print("   sweep %d: %d block pairs rotated (%d with a big rotation); active visits per block: %s" % (sweep, cnt, nbig, act_blocks[:NB].tolist()))
        if cnt == 0 or nbig == 0:
            return sweep + 1, W, active_hist
    return -1, W, active_hist


if __name__ == '__main__':
    path = sys.argv[1]
    rho = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-2
    nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    d = np.load(path)
    for k in list(d.files)[:nmax]:
        W = d[k]
        print("== %s %s rho %g" % (k, W.shape, rho))
        sw, Wf, _ = jacobi_trace(W, rho, name=k)
        s = np.sort(np.linalg.norm(Wf, axis=1))[::-1]
        ref = np.linalg.svd(W, compute_uv=False)
        print("   -> %d sweeps; sigma vs LAPACK max abs err / sigma_max %.1e" % (sw, np.abs(s - ref[:len(s)]).max() / ref[0]))
