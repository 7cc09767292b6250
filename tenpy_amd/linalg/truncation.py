"""Truncation of Schmidt spectra: host decision logic on the (small) singular value vector.

Mirrors ``tenpy/linalg/truncation.py`` (``TruncationError`` :56, ``truncate`` :146, ``svd_theta`` :258).
The SVD itself and the projection of U / VH run on the device; only ``S`` (<= d*chi doubles) is on
the host, as in SURVEY 2.3 K11.
"""
import warnings

import numpy as np

from . import np_conserved as npc

__all__ = ['TruncationError', 'truncate', 'svd_theta']


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
