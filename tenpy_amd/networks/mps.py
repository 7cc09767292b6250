"""Finite MPS of device-resident site tensors for the stand-alone drivers (``algorithms/dmrg.py``, ``algorithms/tebd.py``)
on boxes without TeNPy.  It holds what those two drivers touch -- ``from_product_state``, ``get_B`` / ``set_B`` /
``get_theta`` / ``set_SL`` / ``set_SR`` with the reference's leg labels ``('vL', 'p', 'vR')`` and canonical-form convention
``B = S**nuL  Gamma  S**nuR`` ('A' = (1,0), 'B' = (0,1), 'Th' = (1,1)), ``expectation_value``, ``entanglement_entropy``.
With TeNPy installed, its own 7.6 kLoC ``MPS`` class (infinite systems, canonical forms, correlation functions, ...)
runs unchanged on the mirror (``tenpy_amd/install.py``).
"""
import numpy as np

from ..linalg import np_conserved as npc
from ..linalg.charges import LegCharge

__all__ = ['MPS']

_FORMS = {'A': (1., 0.), 'B': (0., 1.), 'C': (0.5, 0.5), 'G': (0., 0.), 'Th': (1., 1.), None: None}


class MPS:
    bc = 'finite'
    finite = True

    def __init__(self, p_legs, Bs, SVs, form='B'):
        self.p_legs = list(p_legs)          # physical leg of each site
        self.L = len(Bs)
        self._B = list(Bs)
        self._S = [np.asarray(s, dtype=np.float64) for s in SVs]   # L+1 Schmidt spectra, host
        self.form = [_FORMS[form]] * self.L
        self.chinfo = Bs[0].chinfo
        self.dtype = Bs[0].dtype
        self.norm = 1.

    @classmethod
    def from_product_state(cls, p_legs, p_state, dtype=np.float64):
        """Product state; ``p_state[i]`` is the flat physical index occupied on site i."""
        chinfo = p_legs[0].chinfo
        Bs = []
        q_left = chinfo.make_valid()
        for leg_p, occupied in zip(p_legs, p_state):
            qi, _ = leg_p.get_qindex(int(occupied))
            q_right = chinfo.make_valid(q_left + leg_p.get_charge(qi))
            vL = LegCharge.from_qflat(chinfo, [q_left], qconj=+1)
            vR = LegCharge.from_qflat(chinfo, [q_right], qconj=-1)
            dense = np.zeros((1, leg_p.ind_len, 1), dtype=dtype)
            dense[0, int(occupied), 0] = 1.
            Bs.append(npc.Array.from_ndarray(dense, [vL, leg_p, vR], dtype=dtype, labels=['vL', 'p', 'vR']))
            q_left = q_right
        return cls(p_legs, Bs, [np.ones(1)] * (len(p_legs) + 1), form='B')

    @property
    def chi(self):
        return [len(s) for s in self._S[1:-1]]

    def get_SL(self, i):
        return self._S[i]

    def get_SR(self, i):
        return self._S[i + 1]

    def set_SL(self, i, S):
        self._S[i] = np.asarray(S)

    def set_SR(self, i, S):
        self._S[i + 1] = np.asarray(S)

    def set_B(self, i, B, form='B'):
        self._B[i] = B.transpose(['vL', 'p', 'vR']) if B._labels != ['vL', 'p', 'vR'] else B
        self.form[i] = _FORMS[form] if not isinstance(form, tuple) else form

    @staticmethod
    def _scaled(B, S, power, axis):
        if power == 0.:
            return B
        return B.scale_axis(S if power == 1. else S**power, axis)

    def get_B(self, i, form='B', copy=False):
        """Site tensor converted to ``form``; ``form=None`` returns the stored tensor."""
        want = _FORMS[form] if not isinstance(form, tuple) else form
        B = self._B[i]
        if want is not None and want != self.form[i]:
            have = self.form[i]
            B = self._scaled(B, self._S[i], want[0] - have[0], 'vL')
            B = self._scaled(B, self._S[i + 1], want[1] - have[1], 'vR')
        elif copy:
            B = B.copy(deep=True)
        return B

    def get_theta(self, i, n=2, formL=1., formR=1.):
        """Two-site wave function with labels ``'vL', 'p0', 'p1', 'vR'`` (reference mps.py:3041)."""
        assert n == 2
        i1 = i + 1
        B0 = self._scaled(self._B[i], self._S[i], formL - self.form[i][0], 'vL')
        B0 = self._scaled(B0, self._S[i1], 1. - self.form[i][1] - self.form[i1][0], 'vR')
        B1 = self._scaled(self._B[i1], self._S[i1 + 1], formR - self.form[i1][1], 'vR')
        return npc.tensordot(B0.replace_label('p', 'p0'), B1.replace_label('p', 'p1'), axes=['vR', 'vL'])

    def expectation_value(self, op, sites=None):
        """``<psi| op_i |psi>`` for every site in ``sites`` (default: all): ``op`` is a dense (d, d) host matrix [p, p*] (must
        conserve the charges) or a device Array with labels 'p', 'p*' (reference ``MPS.expectation_value`` for one-site
        operators, mps.py).  One tensordot and one inner product per site on the device."""
        sites = range(self.L) if sites is None else sites
        res = []
        for i in sites:
            th = self.get_B(i, 'Th')
            if isinstance(op, npc.Array):
                O = op
            else:
                leg = th.get_leg('p')
                O = npc.Array.from_ndarray(np.asarray(op), [leg, leg.conj()], labels=['p', 'p*'])
            C = npc.tensordot(O, th, axes=['p*', 'p'])
            res.append(npc.inner(th, C, axes='labels', do_conj=True))
        res = np.array(res)
        return np.real_if_close(res)

    def copy(self):
        """Copy sharing the (immutable) tensors: the engines replace tensors, they never modify them in place."""
        res = MPS(self.p_legs, list(self._B), list(self._S), form='B')
        res.form = list(self.form)
        res.norm = self.norm
        return res

    def entanglement_entropy(self):
        res = []
        for s in self._S[1:-1]:
            p = s[s > 1e-30]**2
            res.append(float(-np.sum(p * np.log(p))))
        return np.array(res)

    def norm_test(self):
        """<psi|psi> by contracting the transfer matrices (device tensordots)."""
        B = self.get_B(0, 'B')
        E = npc.tensordot(B.conj(), B, axes=(['vL*', 'p*'], ['vL', 'p']))
        for i in range(1, self.L):
            B = self.get_B(i, 'B')
            E = npc.tensordot(B.conj(), npc.tensordot(E, B, axes=['vR', 'vL']), axes=(['vL*', 'p*'], ['vR*', 'p']))
        return E.to_ndarray().reshape(-1)[0]
