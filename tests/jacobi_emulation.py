"""TEST INFRASTRUCTURE: numpy emulation of the device block-Jacobi SVD iteration (tenpy_amd/csrc/tpa_svd.hip:
8-row blocks, 16 x 16 Gram per block pair recomputed from the data every round, in-LDS two-sided Jacobi on the Gram
with the Hestenes angle, cross-only rounds after the first round of a sweep, rotation counter as convergence test).
It exists to pin the STOPPING RULE on the CPU: tests/test_svd_rule.py runs it on matrices on which the rule without the
null-row cut never terminates (found on the GPU: SVDs of the subspace expansion).  Not used by the product."""
import numpy as np
EPS = 2.220446049250313e-16
BRJ, TRJ = 8, 16

NEW_BIG_RULE = True     # False: the rule of rounds 2-4 (scaled by the floor alone)
NULL_ROW_CUT = 1.0e-60      # svd_needs_rotation: |row|^2 below this fraction of the partner's -> zero row


PRED_BOOST = 1.0          # experiment knobs of round 5 (the device code has neither): a prediction floor^2 = PRED_BOOST x the rotation floor^2
PRED_CAP = float('inf')   # and a cap on the cos^2 a pair below the floor is let go with.  Emulated on chi = 2048 blocks: no sweep saved.


def big_rotation(a, b, g2, floor2, predict=1e-7):
    """svd_big_rotation of tpa_svd.hip, operation for operation (NEW_BIG_RULE False: the rule of rounds 2-4, scaled by the floor alone)."""
    mn, mx0 = min(a, b), max(a, b)
    if not NEW_BIG_RULE:
        return g2 > predict * predict * mn * max(mx0, floor2)
    pf2 = floor2 * PRED_BOOST
    if mx0 >= pf2:
        return g2 > predict * predict * mn * mx0
    return g2 > mn * min(predict * predict * np.sqrt(mx0 * pf2), PRED_CAP * mx0)


def needs(a, b, g2, tol, floor2):
    """svd_needs_rotation of tpa_svd.hip, operation for operation."""
    if not (a > 0.0) or not (b > 0.0):
        return False
    mn = min(a, b)
    mx0 = max(a, b)
    if mn < NULL_ROW_CUT * mx0:
        return False
    mx = max(mx0, floor2)
    return g2 > tol * tol * mn * mx

def block_pair_of(R, pair, rnd):
    NB = (R + BRJ - 1) // BRJ
    NBp = (NB + 1) // 2 * 2
    mod = NBp - 1
    r = rnd % mod if mod > 0 else 0
    if pair == 0:
        bi, bj = NBp - 1, r
    else:
        bi, bj = (r + pair) % mod, (r - pair + mod) % mod
    if bi > bj:
        bi, bj = bj, bi
    return bi, bj, NB

def local_solve(Sm, Q, full_local, tol, floor2, local_sweeps=1):
    n_local = TRJ - 1 if full_local else BRJ
    for sw in range(local_sweeps):
        rotated = False
        for rr in range(n_local):
            part = np.zeros(TRJ, int); cs = np.ones(TRJ); cp = np.zeros(TRJ)
            rot_now = False
            for i in range(TRJ):
                if full_local:
                    if i == TRJ - 1: pi = rr
                    elif i == rr: pi = TRJ - 1
                    else: pi = (2 * rr - i + 2 * (TRJ - 1)) % (TRJ - 1)
                else:
                    pi = (BRJ + ((i + rr) & (BRJ - 1))) if i < BRJ else (((i - BRJ) - rr) & (BRJ - 1))
                p, q = (i, pi) if i < pi else (pi, i)
                a, b, g = Sm[p, p], Sm[q, q], Sm[p, q]
                c, s = 1.0, 0.0
                if needs(a, b, g * g, tol, floor2):
                    zeta = (b - a) / (2.0 * g)
                    h = np.sqrt(zeta * zeta + 1.0)
                    t = np.copysign(1.0, zeta) / (abs(zeta) + h)
                    c = 1.0 / np.sqrt(t * t + 1.0)
                    s = c * t
                    rotated = True; rot_now = True
                part[i] = pi; cs[i] = c; cp[i] = -s if i == p else s
            if not rot_now:
                continue
            Sn = cs[:, None] * Sm + cp[:, None] * Sm[part, :]
            Qn = cs[:, None] * Q + cp[:, None] * Q[part, :]
            Sm[:] = Sn; Q[:] = Qn
            Sn = cs[None, :] * Sm + cp[None, :] * Sm[:, part]
            Sm[:] = Sn
        if not rotated:
            break

def jacobi(A, rho=1e-6, max_sweeps=80, cross_only=True, verbose=False, predict=None):
    """Returns (sweeps, W).  ``predict``: if set (e.g. 1e-8), a sweep in which no checked pair had a scaled cosine above
    it ends the iteration without the verification sweep (quadratic convergence: the rotations of that sweep leave
    cosines ~ predict**2); this is the proposed `tpa_svd_set_algorithm` bit 10, see DESIGN.md section 7."""
    m, n = A.shape
    W = (A if m <= n else A.T).copy()
    R, L = W.shape
    fro2 = float((A * A).sum())
    tol = EPS * np.sqrt(L); floor2 = rho * rho * fro2
    NB = (R + 7) // 8; NBp = (NB + 1) // 2 * 2
    rounds = max(NBp - 1, 1)
    if R < 2:
        return 0, W
    for sweep in range(max_sweeps):
        cnt = 0
        big = 0
        for r in range(rounds):
            full_local = 0 if (cross_only and r > 0) else 1
            for pair in range(NBp // 2):
                bi, bj, NB_ = block_pair_of(R, pair, r)
                rows = []
                for t in range(TRJ):
                    b = bi if t < BRJ else bj
                    rr_ = b * BRJ + (t % BRJ)
                    rows.append(rr_ if (b < NB_ and rr_ < R) else -1)
                X = np.zeros((TRJ, L))
                for t, rw in enumerate(rows):
                    if rw >= 0: X[t] = W[rw]
                Sm = X @ X.T
                flag = False
                flag_big = False
                for ei in range(TRJ):
                    for ej in range(TRJ):
                        rel = (ei < ej) if full_local else (ei < BRJ and ej >= BRJ)
                        if rel and needs(Sm[ei, ei], Sm[ej, ej], Sm[ei, ej] ** 2, tol, floor2):
                            flag = True
                            if predict is not None:
                                a_, b_ = Sm[ei, ei], Sm[ej, ej]
                                # svd_big_rotation of tpa_svd.hip (round 5): the rotation leaves a cosine of ~cos^2, and that must
                                # meet the pair's OWN stopping rule -- for a pair below the floor the scale is sqrt(mx * floor2), not
                                # floor2 (with floor2 such pairs were never "big": the iteration could stop on cosines of O(0.1))
                                if big_rotation(a_, b_, Sm[ei, ej] ** 2, floor2, predict):
                                    flag_big = True
                if not flag:
                    continue
                cnt += 1
                big += 1 if flag_big else 0
                Q = np.eye(TRJ)
                local_solve(Sm, Q, full_local, tol, floor2)
                Xn = Q @ X
                for t, rw in enumerate(rows):
                    if rw >= 0: W[rw] = Xn[t]
        if verbose:
            print(sweep, cnt, np.sort(np.linalg.norm(W, axis=1))[::-1][:8])
        if cnt == 0 or (predict is not None and big == 0):
            return sweep + 1, W
    return -1, W


# ======================================================================================================================
# Round 6: emulation of the DEFAULT device path -- Gram-only sweeps on 32-row blocks (csrc/tpa_svd_b32.inc), both stopping rules
# (floor on the larger row: rounds 1 - 5; on the smaller row: round 6, csrc/tpa_svd.hip::svd_needs_rotation), the activity-driven
# schedule, and the ordered clean-up of linalg/_svd_warm.py::ordered_rows.  Vectorised numpy; used by tests/test_svd_highprec.py.
# ======================================================================================================================
BB32, TB32 = 32, 64


def needs32(a, b, g2, tol, floor2, on_min):
    mn, mx0 = np.minimum(a, b), np.maximum(a, b)
    ok = (a > 0) & (b > 0) & ~(mn < NULL_ROW_CUT * mx0)
    if on_min:
        return ok & (g2 > tol * tol * mx0 * np.maximum(mn, floor2))
    return ok & (g2 > tol * tol * mn * np.maximum(mx0, floor2))


def big32(a, b, g2, floor2, on_min):
    mn, mx0 = np.minimum(a, b), np.maximum(a, b)
    if on_min:
        return np.where(mn >= floor2, g2 > 1e-14 * mn * mx0, g2 > 1e-14 * mx0 * np.sqrt(mn * floor2))
    return np.where(mx0 >= floor2, g2 > 1e-14 * mn * mx0, g2 > 1e-14 * mn * np.sqrt(mx0 * floor2))


def _pair_of32(R, pair, rnd):
    NB = (R + BB32 - 1) // BB32
    NBp = (NB + 1) // 2 * 2
    mod = NBp - 1
    r = rnd % mod if mod > 0 else 0
    bi, bj = (NBp - 1, r) if pair == 0 else ((r + pair) % mod, (r - pair + mod) % mod)
    return (bi, bj) if bi < bj else (bj, bi)


def _local_pairs32(rr, full_local):
    a = np.arange(BB32)
    if not full_local:
        return a, BB32 + ((a + rr) & (BB32 - 1))
    x = np.where(a == 0, TB32 - 1, (rr + a) % (TB32 - 1))
    y = np.where(a == 0, rr, (rr - a + (TB32 - 1)) % (TB32 - 1))
    return np.minimum(x, y), np.maximum(x, y)


def _solve_pair32(Sm, full_local, tol, floor2, on_min):
    if full_local:
        iu = np.triu_indices(TB32, 1)
    else:
        iu = np.nonzero((np.arange(TB32) < BB32)[:, None] & (np.arange(TB32) >= BB32)[None, :])
    d = np.diag(Sm)
    if not needs32(d[iu[0]], d[iu[1]], Sm[iu] ** 2, tol, floor2, on_min).any():
        return None
    Q = np.eye(TB32)
    Sm = Sm.copy()
    for rr in range(TB32 - 1 if full_local else BB32):
        p, q = _local_pairs32(rr, full_local)
        al, be, ga = Sm[p, p], Sm[q, q], Sm[p, q]
        nr = needs32(al, be, ga * ga, tol, floor2, on_min)
        if not nr.any():
            continue
        with np.errstate(all='ignore'):
            zeta = (be - al) / (2.0 * ga)
            t = np.copysign(1.0, zeta) / (np.abs(zeta) + np.sqrt(zeta * zeta + 1.0))
            c = 1.0 / np.sqrt(t * t + 1.0)
            s = c * t
        c, s = np.where(nr, c, 1.0), np.where(nr, s, 0.0)
        J = np.eye(TB32)
        J[p, p], J[q, q], J[p, q], J[q, p] = c, c, -s, s
        Sm = J @ Sm @ J.T
        Q = J @ Q
    return Q


def jacobi_b32(W0, rho=1e-2, on_min=True, max_sweeps=40):
    """One-sided block Jacobi on the ROWS of ``W0`` as the device runs it by default: per sweep one exact Gram matrix, the activity of
    the block pairs on it, the first round of the round-robin schedule (all local pairs) + perfect matchings of the remaining active
    block pairs, one product with the accumulated transform; the sweep that starts without big pairs is the last.
    Returns ``(sweeps, rounds, W, G)``: ``W = G W0`` with ``G`` orthogonal (accumulated rotations)."""
    W = W0.copy()
    R, L = W.shape
    fro2 = float((W * W).sum())
    tol, floor2 = EPS * np.sqrt(L), rho * rho * fro2
    NB = (R + BB32 - 1) // BB32
    NBp = (NB + 1) // 2 * 2
    Gtot = np.eye(R)
    rounds = 0
    for sweep in range(max_sweeps):
        S = W @ W.T
        d = np.diag(S)
        nd = needs32(d[:, None], d[None, :], S * S, tol, floor2, on_min)
        np.fill_diagonal(nd, False)
        if not nd.any():
            return sweep, rounds, W, Gtot
        any_big = bool((nd & big32(d[:, None], d[None, :], S * S, floor2, on_min)).any())
        pad = NBp * BB32 - R
        act = np.pad(nd, ((0, pad), (0, pad))).reshape(NBp, BB32, NBp, BB32).any(axis=(1, 3))
        sched = [[_pair_of32(R, p, 0) for p in range(NBp // 2)]]
        A = act.copy()
        np.fill_diagonal(A, False)
        for (i, j) in sched[0]:
            A[i, j] = A[j, i] = False
        while A.any():
            deg = A.sum(1)
            free = np.ones(NBp, bool)
            m = []
            for v in np.argsort(-deg, kind='stable'):
                if not free[v] or deg[v] == 0:
                    continue
                cand = np.nonzero(A[v] & free)[0]
                cand = cand[cand != v]
                if len(cand) == 0:
                    continue
                w = cand[np.argmax(deg[cand])]
                m.append((min(v, w), max(v, w)))
                free[v] = free[w] = False
                A[v, w] = A[w, v] = False
            rest = np.nonzero(free)[0]
            m += [(rest[k], rest[k + 1]) for k in range(0, len(rest) - 1, 2)]
            sched.append(m)
        Qtot = np.eye(R)
        for r, m in enumerate(sched):
            for (bi, bj) in m:
                idx = np.concatenate([np.arange(bi * BB32, bi * BB32 + BB32), np.arange(bj * BB32, bj * BB32 + BB32)])
                ok = idx < R
                gi = np.where(ok, idx, 0)
                Q = _solve_pair32(S[np.ix_(gi, gi)] * np.outer(ok, ok), r == 0, tol, floor2, on_min)
                if Q is None:
                    continue
                rows, Qs = gi[ok], Q[np.ix_(ok, ok)]
                S[rows, :] = Qs @ S[rows, :]
                S[:, rows] = S[:, rows] @ Qs.T
                Qtot[rows, :] = Qs @ Qtot[rows, :]
        W = Qtot @ W
        Gtot = Qtot @ Gtot
        rounds += len(sched)
        if not any_big:
            return sweep + 1, rounds, W, Gtot
    return -1, rounds, W, Gtot


def ordered_cleanup(V0, iterations=6):
    """``ordered_rows`` of linalg/_svd_warm.py on unit rows sorted by descending weight."""
    T = V0.copy()
    for _ in range(iterations):
        G = T @ T.T
        N = np.tril(G, -1) + np.diag((np.diag(G) - 1.0) / 2.0)
        T = T - N @ T
    return T


def svd_rows_emulated(W0, rho=1e-2, on_min=True, iterations=6):
    """``W0 = Gt^T diag(s) V`` through the emulated iteration + the ordered clean-up: returns ``(s, V, Gt, sweeps, rounds)`` with the
    rows sorted by descending ``s``; ``Gt`` orthogonal to rounding, ``V`` orthonormal after the clean-up."""
    sweeps, rounds, W, G = jacobi_b32(W0, rho, on_min)
    s = np.linalg.norm(W, axis=1)
    order = np.argsort(-s, kind='stable')
    s, W, G = s[order], W[order], G[order]
    nz = s > 1e-15 * np.linalg.norm(s)
    V = np.zeros_like(W)
    V[nz] = ordered_cleanup(W[nz] / s[nz, None], iterations) if on_min else _lowdin(W[nz] / s[nz, None], iterations)
    return s, V, G, sweeps, rounds


def _lowdin(V0, iterations):
    T = V0.copy()
    for _ in range(iterations):
        T = 1.5 * T - 0.5 * (T @ T.T) @ T
    return T
