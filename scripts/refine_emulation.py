"""TEST / MEASUREMENT INFRASTRUCTURE (numpy): the refinement end game of the block SVD (csrc/tpa_svd_refine.inc) on the host, operation
for operation what the host loop of svd_run does -- simultaneous Hestenes rotations from one exact Gram matrix, row-sum bound of |K|,
Newton-Schulz plan, predicted convergence -- started after 0 .. 3 cyclic one-sided Jacobi sweeps on the pivoted-QR factor of a block of
the saturated chi = 2048 theta.  Usage: first ``python scripts/refine_emulation.py prepare 4`` (writes the sweep states of block 4 to
/tmp), then ``python scripts/refine_emulation.py 4 3 2 1 0``.  Not used by the product."""
import numpy as np, sys, os
EPS = 2.220446049250313e-16
def prepare(blk):
    import scipy.linalg as sla
    A = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'theta_chi2048_sat.npz'))['arr_%s' % blk]
    m, n = A.shape
    fro = np.linalg.norm(A)
    _, R0, _ = sla.qr(A if m >= n else A.T, pivoting=True, mode='economic')
    dn = np.linalg.norm(R0, axis=1)
    tail = np.sqrt(np.cumsum((dn ** 2)[::-1])[::-1])
    r = int(np.sum(tail > 1e-15 * fro))
    X = R0[:r].copy()
    L = X.shape[1]
    tol = EPS * np.sqrt(L); floor2 = (1e-6 * fro) ** 2
    out = {'X0': X.copy(), 'Sref': np.linalg.svd(A, compute_uv=False), 'fro': fro}
    for k in range(1, 5):          # cyclic one-sided Jacobi, circle-method rounds, vectorised over the disjoint pairs of a round
        Rp = r + (r & 1)
        for rnd in range(Rp - 1):
            a = np.empty(Rp // 2, int); b = np.empty(Rp // 2, int)
            a[0] = Rp - 1; b[0] = rnd
            kk = np.arange(1, Rp // 2)
            a[1:] = (rnd + kk) % (Rp - 1); b[1:] = (rnd - kk) % (Rp - 1)
            ok = (a < r) & (b < r); a, b = a[ok], b[ok]
            p = np.minimum(a, b); q = np.maximum(a, b)
            xp, xq = X[p], X[q]
            al = np.einsum('ij,ij->i', xp, xp); be = np.einsum('ij,ij->i', xq, xq); g = np.einsum('ij,ij->i', xp, xq)
            need = (g * g > tol * tol * np.minimum(al, be) * np.maximum(np.maximum(al, be), floor2)) & (al > 0) & (be > 0)
            zeta = (be - al) / (2 * np.where(need, g, 1.0))
            t = np.sign(zeta) / (np.abs(zeta) + np.sqrt(1 + zeta * zeta)); t = np.where(zeta == 0, 1.0, t)
            c = 1 / np.sqrt(1 + t * t); s = c * t
            c = np.where(need, c, 1.0); s = np.where(need, s, 0.0)
            X[p] = c[:, None] * xp - s[:, None] * xq
            X[q] = s[:, None] * xp + c[:, None] * xq
        out['X%d' % k] = X.copy()
    np.savez('/tmp/refine_sweeps_%s.npz' % blk, **out)


if sys.argv[1] == 'prepare':
    prepare(sys.argv[2])
    sys.exit(0)
blk = sys.argv[1]
d = np.load('/tmp/refine_sweeps_%s.npz' % blk)
Sref = d['Sref']; fro = float(d['fro'])
def ns_plan(kinf):
    scale = 1.0
    if not kinf * kinf > 1e-17: return 0, 1.0
    smax = np.sqrt(1 + kinf * kinf)
    if smax > 1.1: scale = 1.1 / smax
    lo, hi, n = scale, scale * smax, 0
    while n < 24 and (abs(1 - lo * lo) > 2e-16 or abs(1 - hi * hi) > 2e-16):
        lo = lo * (3 - lo * lo) / 2; hi = hi * (3 - hi * hi) / 2; n += 1
    return n, scale
for start in sys.argv[2:]:
    X = d['X' + start].copy(); r, L = X.shape
    tol = EPS * np.sqrt(L); floor2 = (1e-6 * fro) ** 2
    print("== start after", start, "sweeps")
    tot = 0
    for it in range(16):
        S = X @ X.T; dd = np.diag(S).copy()
        mn = np.minimum.outer(dd, dd); mx = np.maximum(np.maximum.outer(dd, dd), floor2)
        need = (S * S > tol * tol * mn * mx); np.fill_diagonal(need, False)
        big = (S * S > 1e-14 * mn * mx); np.fill_diagonal(big, False)
        n_need = need.sum() // 2; n_big = (need & big).sum() // 2
        if n_need == 0: print("  converged at it", it); break
        Ss = np.where(need, S, 1.0)
        zeta = (dd[None, :] - dd[:, None]) / (2 * Ss)
        t = np.sign(zeta) / (np.abs(zeta) + np.sqrt(1 + zeta * zeta)); t = np.where(zeta == 0, 1.0, t); t = np.where(need, t, 0.0)
        K = np.triu(-t, 1); K = K - K.T
        kinf = np.abs(K).sum(axis=1).max(); k2 = np.linalg.norm(K, 2)
        ns, scale = ns_plan(kinf)
        Q = (np.eye(r) + K)
        e_hist = []
        for k in range(ns):
            s_ = scale if k == 0 else 1.0
            T = Q @ Q.T
            e_hist.append(np.abs(s_ * s_ * T - np.eye(r)).max())
            Q = (1.5 * s_ * np.eye(r) - 0.5 * s_ ** 3 * T) @ Q
        efin = np.abs(Q @ Q.T - np.eye(r)).max()
        tot += ns
        X = Q @ X
        print("  it %d need %d big %d kinf %.2e k2 %.2e ns %d e_last %.1e e_fin %.1e" % (it, n_need, n_big, kinf, k2, ns, e_hist[-1] if e_hist else 0, efin))
        if n_big == 0: print("  predicted convergence after it", it); break
    sv = np.sort(np.linalg.norm(X, axis=1))[::-1]
    print("  total ns", tot, "sv err %.2e" % (np.abs(sv - Sref[:r]).max() / Sref[0]), "orth cos max", end=' ')
    S = X @ X.T; dd = np.diag(S); c = np.abs(S) / np.sqrt(np.minimum.outer(dd, dd) * np.maximum(np.maximum.outer(dd, dd), floor2)); np.fill_diagonal(c, 0); print("%.2e" % c.max())
