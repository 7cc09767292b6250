"""TEST INFRASTRUCTURE: singular values in EXTENDED precision, the arbiter between LAPACK (the reference's ``svd_flat``,
np_conserved.py:4970 via svd_robust.py:36) and the device block SVD for singular values far below ``sigma_max`` -- where LAPACK
itself only promises ``eps * sigma_max`` ABSOLUTE accuracy, so that it cannot judge an algorithm of the same class.

``sv_reference(A)``: fp64 LAPACK vectors as a preconditioner, ``M = U0^T A V0`` formed in ``np.longdouble`` (64-bit mantissa,
eps 1.1e-19), one-sided Jacobi on the rows of the nearly diagonal ``M`` in ``np.longdouble``.  The fp64 factors are orthogonal to
~1e-15 only, but that is a MULTIPLICATIVE perturbation: it moves every singular value by a relative 1e-15, tiny ones included
(Ostrowski); what limits the reference is the rounding of the extended-precision products, ~1e-19 sigma_max absolute -- a factor
~1000 below the errors of any fp64 algorithm.  Checked against ``mpmath`` (200 bits) in tests/test_svd_highprec.py.
Not used by the product."""
import numpy as np

LD = np.longdouble


def jacobi_rows_ld(M, max_sweeps=8):
    """Row norms (descending) after one-sided Jacobi (Hestenes) on the rows of ``M`` in extended precision."""
    M = np.array(M, dtype=LD)
    k = M.shape[0]
    kp = k + (k & 1)
    tol = LD(2e-19) * np.sqrt(LD(M.shape[1]))
    players = np.arange(kp)
    for _ in range(max_sweeps):
        rotated = 0
        for _r in range(kp - 1):
            a, b = players[:kp // 2], players[kp // 2:][::-1]
            ok = (a < k) & (b < k)
            p, q = np.minimum(a[ok], b[ok]), np.maximum(a[ok], b[ok])
            X, Y = M[p], M[q]
            al, be, ga = np.einsum('ij,ij->i', X, X), np.einsum('ij,ij->i', Y, Y), np.einsum('ij,ij->i', X, Y)
            need = (al > 0) & (be > 0) & (np.abs(ga) > tol * np.sqrt(al * be))
            if need.any():
                rotated += int(need.sum())
                g = np.where(need, ga, LD(1))
                zeta = (be - al) / (2 * g)
                t = np.where(zeta >= 0, LD(1), LD(-1)) / (np.abs(zeta) + np.sqrt(1 + zeta * zeta))
                c = 1 / np.sqrt(1 + t * t)
                s = c * t
                c, s = np.where(need, c, LD(1)), np.where(need, s, LD(0))
                M[p] = c[:, None] * X - s[:, None] * Y
                M[q] = s[:, None] * X + c[:, None] * Y
            players = np.concatenate([players[:1], players[-1:], players[1:-1]])
        if rotated == 0:
            break
    return np.sort(np.sqrt(np.einsum('ij,ij->i', M, M)))[::-1]


def sv_reference(A):
    """Singular values of the fp64 matrix ``A`` (descending, ``np.longdouble``), absolute accuracy ~1e-19 sigma_max."""
    A = np.asarray(A, dtype=np.float64)
    u, _, vh = np.linalg.svd(A, full_matrices=False)
    M = (u.T.astype(LD) @ A.astype(LD)) @ vh.T.astype(LD)
    return jacobi_rows_ld(M)


def graded_block(rng, m, n, r, decades=14.5):
    """``m x n`` block of numerical rank ``r`` with singular values spread evenly over ``decades`` decades: what the charge blocks
    of a saturated DMRG wave function look like (the chi = 2048 theta of the bench: 14.5 decades down to rounding level)."""
    u, _ = np.linalg.qr(rng.standard_normal((m, r)))
    v, _ = np.linalg.qr(rng.standard_normal((n, r)))
    return (u * np.logspace(0, -decades, r)) @ v.T


def rel_err_by_decade(s, ref, lowest=15):
    """``{decade d: max |s_i - ref_i| / ref_i over the values with 10^-(d+1) < ref_i / ref_0 <= 10^-d}``, d = 0 .. lowest - 1."""
    ref = np.asarray(ref, dtype=LD)
    n = min(len(s), len(ref))
    s, ref = np.sort(np.asarray(s))[::-1][:n].astype(LD), ref[:n]
    pos = ref > 0
    with np.errstate(all='ignore'):
        dec = np.where(pos, np.floor(-np.log10(np.where(pos, ref / ref[0], LD(1))).astype(np.float64) + 1e-12), 99).astype(int)
    out = {}
    for d in range(lowest):
        sel = dec == d
        if sel.any():
            out[d] = float(np.max(np.abs(s[sel] - ref[sel]) / ref[sel]))
    return out
