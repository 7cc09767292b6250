"""Truncation of Schmidt spectra: host decision logic on the (small) singular value vector.

Mirrors ``tenpy/linalg/truncation.py`` (``TruncationError`` :56, ``truncate`` :146, ``svd_theta`` :258).
The SVD itself and the projection of U / VH run on the device; only ``S`` (<= d*chi doubles) is on
the host, as in SURVEY 2.3 K11.
"""
import warnings

import numpy as np

from . import np_conserved as npc

__all__ = ['TruncationError', 'truncate', 'svd_theta', 'decompose_theta_qr_based']


class TruncationError:
    """``eps`` = sum of discarded squared Schmidt values, ``ov`` = lower bound of the overlap."""

    def __init__(self, eps=0., ov=1.):
        self.eps = eps
        self.ov = ov

    def copy(self):
        return TruncationError(self.eps, self.ov)

    @classmethod
    def from_norm(cls, norm_new, norm_old=1.):
        eps = 1. - norm_new**2 / norm_old**2
        return cls(eps, 1. - 2. * eps)

    @classmethod
    def from_S(cls, S_discarded, norm_old=None):
        eps = np.sum(np.square(S_discarded))
        if norm_old:
            eps /= norm_old * norm_old
        return cls(eps, 1. - 2. * eps)

    def __add__(self, other):
        return TruncationError(self.eps + other.eps, self.ov * other.ov)

    @property
    def ov_err(self):
        return 1. - self.ov

    def __repr__(self):
        if self.eps != 0 or self.ov != 1.:
            return "TruncationError(eps={eps:.4e}, ov={ov:.10f})".format(eps=self.eps, ov=self.ov)
        return "TruncationError()"


def _restrict(allowed, constraint, name):
    """AND a new constraint into the set of allowed cuts, unless nothing would remain."""
    both = allowed & constraint
    if both.any():
        return both
    warnings.warn("truncation: can't satisfy constraint for " + name, stacklevel=3)
    return allowed


def truncate(S, options):
    """Which Schmidt values to keep.  Options ``chi_max`` (100), ``chi_min``, ``degeneracy_tol``,
    ``svd_min`` (1e-14), ``trunc_cut`` (1e-14) with the reference's semantics and priorities
    (truncation.py:146-255).  Returns ``(mask, norm_new, TruncationError)``."""
    get = options.get
    chi_max = get('chi_max', 100)
    chi_min = get('chi_min', None)
    deg_tol = get('degeneracy_tol', None)
    svd_min = get('svd_min', 1.e-14)
    trunc_cut = get('trunc_cut', 1.e-14)
    S = np.asarray(S)
    n = len(S)
    if trunc_cut is not None and trunc_cut >= 1.:
        raise ValueError("trunc_cut >=1.")
    if not np.any(S > 1.e-10):
        warnings.warn("no Schmidt value above 1.e-10", stacklevel=2)
    if np.any(S < -1.e-10):
        warnings.warn("negative Schmidt values!", stacklevel=2)
    safe = np.where(S <= 0., 1.e-100, S)
    logS = np.log(safe)
    order = np.argsort(logS)          # ascending: the candidates to drop come first
    logS = logS[order]
    # allowed[c] : may we drop order[:c] and keep order[c:] ?
    allowed = np.ones(n, dtype=np.bool_)
    if chi_max is not None:
        c = np.zeros(n, dtype=np.bool_)
        c[-chi_max:] = True
        allowed = _restrict(allowed, c, "chi_max")
    if chi_min is not None and chi_min > 1:
        c = np.ones(n, dtype=np.bool_)
        c[-chi_min + 1:] = False
        allowed = _restrict(allowed, c, "chi_min")
    if deg_tol:
        c = np.empty(n, dtype=np.bool_)
        c[0] = True
        c[1:] = (logS[1:] - logS[:-1]) >= deg_tol
        allowed = _restrict(allowed, c, "degeneracy_tol")
    if svd_min is not None:
        allowed = _restrict(allowed, logS >= np.log(svd_min), "svd_min")
    if trunc_cut is not None:
        allowed = _restrict(allowed, np.cumsum(S[order]**2) > trunc_cut * trunc_cut, "trunc_cut")
    cut = int(np.nonzero(allowed)[0][0])
    mask = np.zeros(n, dtype=np.bool_)
    mask[order[cut:]] = True
    norm_new = np.linalg.norm(S[mask])
    return mask, norm_new, TruncationError.from_S(S[~mask])


def svd_theta(theta, trunc_par, qtotal_LR=[None, None], inner_labels=['vR', 'vL']):
    """SVD of the two-site wave function + truncation (reference truncation.py:258-313).

    Returns ``U, S, VH, err, renormalization`` with ``S`` normalised to 1.
    """
    U, S, VH = npc.svd(theta, full_matrices=False, compute_uv=True, qtotal_LR=qtotal_LR, inner_labels=inner_labels)
    renormalization = np.linalg.norm(S)
    S = S / renormalization
    keep, new_norm, err = truncate(S, trunc_par)
    new_len = int(np.sum(keep))
    if new_len * 100 < len(S) and (trunc_par.get('chi_max', 100) is None or new_len != trunc_par.get('chi_max', 100)):
        warnings.warn("Catastrophic reduction in chi: {0:d} -> {1:d}".format(len(S), new_len), stacklevel=2)
    S = S[keep] / new_norm
    renormalization *= new_norm
    if not np.all(keep):
        U.iproject(keep, axes=1)
        VH.iproject(keep, axes=0)
    return U, S, VH, err, renormalization


def svd_theta_batched(thetas, trunc_par, qtotal_LRs=None, inner_labels=['vR', 'vL']):
    """``[svd_theta(theta, trunc_par, q, inner_labels) for ...]`` for the independent two-site wave functions of one Trotter half-step
    (reference ``algorithms/tebd.py:374-414`` loops over ``np.arange(int(odd) % 2, L, 2)``): ONE batched block SVD
    (``np_conserved.svd_batched``), then the reference's truncation per bond (:258-313)."""
    out = []
    for U, S, VH in npc.svd_batched(thetas, qtotal_LRs, inner_labels=inner_labels):
        renormalization = np.linalg.norm(S)
        S = S / renormalization
        keep, new_norm, err = truncate(S, trunc_par)
        new_len = int(np.sum(keep))
        if new_len * 100 < len(S) and (trunc_par.get('chi_max', 100) is None or new_len != trunc_par.get('chi_max', 100)):
            warnings.warn("Catastrophic reduction in chi: {0:d} -> {1:d}".format(len(S), new_len), stacklevel=2)
        S = S[keep] / new_norm
        renormalization *= new_norm
        if not np.all(keep):
            U.iproject(keep, axes=1)
            VH.iproject(keep, axes=0)
        out.append((U, S, VH, err, renormalization))
    return out


# ======================================================================================================
# QR-based decomposition of theta (reference truncation.py:370-711), the variant the reference flags as
# "faster on GPUs" (algorithms/tebd.py:658-661): two tensordots + two block QRs + an SVD (or eigh) of the
# small bond matrix Xi instead of the SVD of the (d chi) x (d chi) theta.
# ======================================================================================================

def _eig_based_svd(A, need_U=True, need_Vd=True, inner_labels=[None, None], trunc_params=None):
    """SVD of a matrix through ``eigh`` of ``A A^dagger`` or ``A^dagger A`` (reference :473).  Only one of U / Vd."""
    return _eig_based_svd_batched([A], need_U, need_Vd, inner_labels, trunc_params)[0]


def _eig_based_svd_batched(As, need_U=True, need_Vd=True, inner_labels=[None, None], trunc_params=None):
    """:func:`_eig_based_svd` for independent matrices (the bond matrices of one Trotter half-step): the Hermitian eigenproblems of
    all of them in ONE batched device call (``np_conserved.eigh_batched``), then the reference's truncation per matrix."""
    if need_U and need_Vd:
        raise NotImplementedError
    for A in As:
        assert A.rank == 2
    Us, Vds = [None] * len(As), [None] * len(As)
    if need_U:
        res = npc.eigh_batched([npc.tensordot(A, A.conj(), [1, 1]) for A in As], sort='>')
        Ss = [np.sqrt(np.abs(L)) for L, _ in res]
        Us = [U.ireplace_label('eig', inner_labels[0]) for _, U in res]
    elif need_Vd:
        res = npc.eigh_batched([npc.tensordot(A.conj(), A, [0, 0]) for A in As], sort='>')
        Ss = [np.sqrt(np.abs(L)) for L, _ in res]
        Vds = [V.iconj().itranspose().ireplace_label('eig*', inner_labels[1]) for _, V in res]
    else:
        A2s = [npc.tensordot(A, A.conj(), [1, 1]) if A.shape[1] >= A.shape[0] else npc.tensordot(A.conj(), A, [0, 0]) for A in As]
        Ss = [np.sqrt(np.abs(L)) for L, _ in npc.eigh_batched(A2s)]
    out = []
    for U, S, Vd in zip(Us, Ss, Vds):
        if trunc_params is not None:
            keep, renormalize, trunc_err = truncate(S, trunc_params)
            S = S[keep] / renormalize
            if need_U:
                U.iproject(keep, 1)
            if need_Vd:
                Vd.iproject(keep, 0)
        else:
            renormalize = np.linalg.norm(S)
            S = S / renormalize
            trunc_err = TruncationError()
        out.append((U, S, Vd, trunc_err, renormalize))
    return out


def _qr_theta_Y0(old_qtotal_L, old_qtotal_R, old_bond_leg, theta, move_right, expand, min_block_increase):
    """Initial guess Y0 for the isometry: theta with the fused leg restricted to the old bond sectors enlarged by
    ``expand`` (the slices of largest norm are kept; reference :370-470).  Legs [vL, (p1.vR)] / [(vL.p0), vR]."""
    assert min_block_increase >= 0 and expand is not None and expand != 0
    Y0 = theta.copy(deep=False)
    Y0.legs = list(theta.legs)
    if move_right:
        Y0.legs[1] = Y0.legs[1].to_LegCharge()
        Y0.ireplace_label('(p1.vR)', 'vR')
        if np.any(np.asarray(old_qtotal_R) != 0):
            Y0 = Y0.gauge_total_charge('vR', old_qtotal_L)
        lab, q_axis = 'vR', 1
    else:
        Y0.legs[0] = Y0.legs[0].to_LegCharge()
        Y0.ireplace_label('(vL.p0)', 'vL')
        if np.any(np.asarray(old_qtotal_L) != 0):
            Y0 = Y0.gauge_total_charge('vL', old_qtotal_R)
        lab, q_axis = 'vL', 0
    Y0._skey = None
    v_old = old_bond_leg if old_bond_leg.is_blocked() else old_bond_leg.sort()[1]
    v_new = Y0.get_leg(lab)
    keep = np.zeros(v_new.ind_len, dtype=bool)
    increase = max(min_block_increase, int(v_old.ind_len * expand // v_new.block_number))
    sizes_old, sizes_new = v_old.get_block_sizes(), v_new.get_block_sizes()
    norms2 = Y0.axis_sqnorms(q_axis)                       # one device pass instead of a norm per block
    have_block = np.zeros(v_new.block_number, dtype=bool)
    have_block[Y0._qdata[:, q_axis]] = True
    j_old = 0
    for j_new in range(v_new.block_number):
        if j_old < v_old.block_number and np.array_equal(v_new.charges[j_new], v_old.charges[j_old]):
            s_new = sizes_old[j_old] + increase
            j_old += 1
        else:
            s_new = increase
        s_new = min(int(s_new), int(sizes_new[j_new]))
        if not have_block[j_new]:
            continue
        start = v_new.slices[j_new]
        nb = norms2[start:v_new.slices[j_new + 1]]
        keep[start + np.argsort(-nb, kind='stable')[:s_new]] = True
    Y0.iproject(keep, lab)
    return Y0


def _decompose_qr_prepare(old_qtotal_L, old_qtotal_R, old_bond_leg, theta, move_right, expand, min_block_increase):
    """First half of :func:`decompose_theta_qr_based`: the two block QRs -> isometries ``A_L``, ``B_R`` and the small bond matrix
    ``Xi`` (reference truncation.py:611-640)."""
    def onto_right(mat, iso):        # mat . iso^dagger : [(vL.p0), (p1.vR)] -> [(vL.p0), vR]
        return npc.tensordot(mat, iso.conj(), ['(p1.vR)', '(p1*.vR*)']).ireplace_label('vL*', 'vR')

    def onto_left(iso, mat):         # iso^dagger . mat : [(vL.p0), (p1.vR)] -> [vL, (p1.vR)]
        return npc.tensordot(iso.conj(), mat, ['(vL*.p0*)', '(vL.p0)']).ireplace_label('vR*', 'vL')

    def left_isometry(mat):          # [(vL.p0), vR] = A . R
        return npc.qr(mat, inner_labels=['vR', 'vL'])

    def right_isometry(mat):         # [vL, (p1.vR)] = R . B   (QR of the transpose)
        q, r = npc.qr(mat.itranspose(['(p1.vR)', 'vL']), inner_labels=['vL', 'vR'], inner_qconj=-1)
        return q.itranspose(['vL', '(p1.vR)']), r.itranspose(['vL', 'vR'])

    guess = _qr_theta_Y0(old_qtotal_L, old_qtotal_R, old_bond_leg, theta, move_right, expand, min_block_increase)
    if move_right:                   # guess spans the left space: first B from it, then A from theta . B^dagger
        B_R, _ = right_isometry(onto_left(guess, theta))
        A_L, Xi = left_isometry(onto_right(theta, B_R))
    else:                            # guess spans the right space
        A_L, _ = left_isometry(onto_right(theta, guess))
        B_R, Xi = right_isometry(onto_left(A_L, theta))
    return A_L, B_R, Xi


def _decompose_qr_finish(theta, A_L, B_R, Xi, U, S, Vd, renormalization, move_right, compute_err, return_both_T):
    """Second half of :func:`decompose_theta_qr_based`: the factors of theta from the decomposed bond matrix (reference :641-711)."""
    want_both = bool(return_both_T or compute_err)

    def left_factor():               # -> (tensor, canonical form)
        if U is not None:
            return npc.tensordot(A_L, U, ['vR', 'vL']), 'A'
        t = npc.tensordot(npc.tensordot(A_L, Xi, ['vR', 'vL']), Vd.conj(), ['vR', 'vR*']).ireplace_label('vL*', 'vR')
        return t.iscale_prefactor(1. / npc.norm(t)), 'Th'

    def right_factor():
        if Vd is not None:
            return npc.tensordot(Vd, B_R, ['vR', 'vL']), 'B'
        t = npc.tensordot(U.conj(), npc.tensordot(Xi, B_R, ['vR', 'vL']), ['vL*', 'vL']).ireplace_label('vR*', 'vL')
        return t.iscale_prefactor(1. / npc.norm(t)), 'Th'

    T_Lc = T_Rc = None
    form = ['A', 'B']
    if move_right or want_both:
        T_Lc, form[0] = left_factor()
    if (not move_right) or want_both:
        T_Rc, form[1] = right_factor()
    trunc_err = TruncationError(np.nan, np.nan)
    if compute_err:                  # || theta / |theta|  -  renormalization / |theta| * T_Lc S T_Rc ||^2
        weighted = T_Lc if 'Th' in form else T_Lc.scale_axis(S, axis='vR')
        scale = 1. / npc.norm(theta)
        residual = theta * scale
        residual.iadd_prefactor_other(-renormalization * scale, npc.tensordot(weighted, T_Rc, ['vR', 'vL']))
        eps = npc.norm(residual)**2
        trunc_err = TruncationError(eps, 1. - 2. * eps)
    if T_Lc is not None:
        T_Lc.ireplace_label('(vL.p0)', '(vL.p)')
    if T_Rc is not None:
        T_Rc.ireplace_label('(p1.vR)', '(p.vR)')
    return T_Lc, S, T_Rc, form, trunc_err, renormalization


def decompose_theta_qr_based(old_qtotal_L, old_qtotal_R, old_bond_leg, theta, move_right, expand, min_block_increase,
                             use_eig_based_svd, trunc_params, compute_err, return_both_T):
    """``theta`` [(vL.p0), (p1.vR)] ~= renormalization * T_Lc . diag(S) . T_Rc without an SVD of theta itself: two block
    QRs give isometries A (left) and B (right) around a small bond matrix ``Xi``, which alone is decomposed (block SVD,
    or ``_eig_based_svd``).  Same arguments and returned tuple ``(T_Lc, S, T_Rc, form, trunc_err, renormalization)`` as the
    reference (truncation.py:533-711); everything is block GEMM / block QR on the device.

    The SVD of the bond matrix runs with the GENERIC stopping rule (``np_conserved.SVD_ABS_FLOOR_GENERIC`` = 0: every returned vector
    converged), not with the engines' absolute floor: this is a library function any caller may use, ``Xi`` is only chi x chi, and the
    measured QR-route numbers are with floor 0.  Only ``svd_theta`` of a full two-site theta inside the DMRG / SVD-based
    TEBD drivers opts into the floor (``svd_hint`` / ``svd_engine_floor``)."""
    A_L, B_R, Xi = _decompose_qr_prepare(old_qtotal_L, old_qtotal_R, old_bond_leg, theta, move_right, expand, min_block_increase)
    if use_eig_based_svd:            # only the factor on the side we move to comes out of the eigen-decomposition
        U, S, Vd, _, renormalization = _eig_based_svd(Xi, need_U=move_right, need_Vd=not move_right,
                                                      inner_labels=['vR', 'vL'], trunc_params=trunc_params)
    else:
        U, S, Vd, _, renormalization = svd_theta(Xi, trunc_params)
    return _decompose_qr_finish(theta, A_L, B_R, Xi, U, S, Vd, renormalization, move_right, compute_err, return_both_T)


def _decompose_qr_prepare_batched(items):
    """:func:`_decompose_qr_prepare` for independent items ``(old_qtotal_L, old_qtotal_R, old_bond_leg, theta, move_right, expand,
    min_block_increase)`` with a common ``move_right``: both block QRs of all items in one batched device call each."""
    move_right = items[0][4]
    assert all(it[4] == move_right for it in items)
    thetas = [it[3] for it in items]
    guesses = [_qr_theta_Y0(*it) for it in items]

    def onto_right(mat, iso):
        return npc.tensordot(mat, iso.conj(), ['(p1.vR)', '(p1*.vR*)']).ireplace_label('vL*', 'vR')

    def onto_left(iso, mat):
        return npc.tensordot(iso.conj(), mat, ['(vL*.p0*)', '(vL.p0)']).ireplace_label('vR*', 'vL')

    def left_isometries(mats):
        return npc.qr_batched(mats, inner_labels=['vR', 'vL'])

    def right_isometries(mats):
        res = npc.qr_batched([m.itranspose(['(p1.vR)', 'vL']) for m in mats], inner_labels=['vL', 'vR'], inner_qconj=-1)
        return [(q.itranspose(['vL', '(p1.vR)']), r.itranspose(['vL', 'vR'])) for q, r in res]

    if move_right:
        B_Rs = [b for b, _ in right_isometries([onto_left(g, t) for g, t in zip(guesses, thetas)])]
        AX = left_isometries([onto_right(t, b) for t, b in zip(thetas, B_Rs)])
        return [(A_L, B_R, Xi) for (A_L, Xi), B_R in zip(AX, B_Rs)]
    A_Ls = [q for q, _ in left_isometries([onto_right(t, g) for t, g in zip(thetas, guesses)])]
    BX = right_isometries([onto_left(A_L, t) for A_L, t in zip(A_Ls, thetas)])
    return [(A_L, B_R, Xi) for A_L, (B_R, Xi) in zip(A_Ls, BX)]


def decompose_theta_qr_based_batched(items, trunc_params, compute_err, return_both_T, use_eig_based_svd=False):
    """:func:`decompose_theta_qr_based` (block SVD of the bond matrix) for several INDEPENDENT two-site wave functions -- the bonds of one
    Trotter half-step -- with the two block QRs and the bond matrices of all of them in ONE batched device call each (``np_conserved.qr_batched``,
    ``svd_theta_batched``, or ``eigh_batched`` with ``use_eig_based_svd``); ``items`` = list of
    ``(old_qtotal_L, old_qtotal_R, old_bond_leg, theta, move_right, expand, min_block_increase)``.  Same results item by item."""
    prep = _decompose_qr_prepare_batched(items)
    if use_eig_based_svd:            # only the factor on the side we move to comes out of the eigen-decompositions
        move_right = items[0][4]
        res = _eig_based_svd_batched([p[2] for p in prep], need_U=move_right, need_Vd=not move_right, inner_labels=['vR', 'vL'],
                                     trunc_params=trunc_params)
    else:
        res = svd_theta_batched([p[2] for p in prep], trunc_params)
    return [_decompose_qr_finish(it[3], A_L, B_R, Xi, U, S, Vd, renorm, it[4], compute_err, return_both_T)
            for it, (A_L, B_R, Xi), (U, S, Vd, _, renorm) in zip(items, prep, res)]
