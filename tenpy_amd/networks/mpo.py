"""Finite MPO and MPO environments for the stand-alone drivers (boxes without TeNPy): ``MPO`` (list of W with labels
``'wL','wR','p','p*'``), ``mpo_from_dense`` (charges of the virtual legs deduced from the entries) and ``MPOEnvironment`` with
cached ``LP`` / ``RP`` (labels ``'vR*','wR','vR'`` and ``'vL','wL','vL*'``; ``_contract_LP`` / ``_contract_RP`` as reference
networks/mpo.py:3087-3105).  Everything is a device tensordot; at chi=2048 all L environments (~100 x 2 x 35 MB) stay resident
in HBM instead of the reference's disk cache.  With TeNPy installed its own classes run on the mirror.
"""
import numpy as np

from ..linalg import np_conserved as npc
from ..linalg.charges import LegCharge

__all__ = ['MPO', 'MPOEnvironment', 'mpo_from_dense']


class MPO:
    bc = 'finite'
    finite = True
    explicit_plus_hc = False

    def __init__(self, p_legs, Ws, IdL=0, IdR=-1):
        self.p_legs = list(p_legs)
        self._W = list(Ws)
        self.L = len(Ws)
        self.IdL, self.IdR = IdL, IdR
        self.chinfo = Ws[0].chinfo
        self.dtype = Ws[0].dtype

    def get_W(self, i):
        return self._W[i]

    @property
    def chi(self):
        return [W.get_leg('wL').ind_len for W in self._W] + [self._W[-1].get_leg('wR').ind_len]


def mpo_from_dense(W_dense_list, p_legs, chinfo, dtype=np.float64, IdL=0, IdR=-1):
    """Build a finite MPO from dense ``W[i]`` of shape (D_l, D_r, d, d).  The charges of the virtual MPO legs are deduced
    from the non-zero entries (every entry must conserve charge) in two passes: forward from the left boundary (index
    ``IdL`` of the first leg = charge 0) and backward from the right boundary (index ``IdR`` of the last leg = charge 0).
    An index that neither pass reaches cannot contribute to any contraction; its entries are dropped.  An index reached
    from one side only is kept: e.g. the "only identities to the right" state on the bond right of the first site when
    no one-site term exists -- the subspace expansion relies on ``W[IdR, IdR] = 1`` being there on every site."""
    L = len(W_dense_list)
    Ws_d = [np.array(Wd) for Wd in W_dense_list]
    nq = chinfo.qnumber
    pqs = [p.to_qflat() * p.qconj for p in p_legs]

    def entries(Wd):
        nz = np.abs(Wd) > 1e-15
        out = {}
        for a, b, s, t in np.argwhere(nz):
            out.setdefault((int(a), int(b)), (int(s), int(t)))       # first non-zero physical entry of each (a, b)
        return out
    ents = [entries(Wd) for Wd in Ws_d]
    # charge rule of an entry: q_a - q_b + p[s] - p[t] = 0
    fq = [np.zeros((Ws_d[0].shape[0], nq), dtype=np.int64)]
    fk = [np.zeros(Ws_d[0].shape[0], dtype=bool)]
    fk[0][IdL % Ws_d[0].shape[0]] = True
    if Ws_d[0].shape[0] == 1:
        fk[0][:] = True
    for i in range(L):
        Dr = Ws_d[i].shape[1]
        q, k = np.zeros((Dr, nq), dtype=np.int64), np.zeros(Dr, dtype=bool)
        for (a, b), (s_, t_) in sorted(ents[i].items()):
            if not fk[i][a]:
                continue
            qb = chinfo.make_valid(fq[i][a] + pqs[i][s_] - pqs[i][t_])
            if k[b] and np.any(q[b] != qb):
                raise ValueError("MPO entry (%d,%d) on site %d violates charge conservation" % (a, b, i))
            q[b], k[b] = qb, True
        fq.append(q)
        fk.append(k)
    bq = [None] * (L + 1)
    bk = [None] * (L + 1)
    Dlast = Ws_d[-1].shape[1]
    bq[L], bk[L] = np.zeros((Dlast, nq), dtype=np.int64), np.zeros(Dlast, dtype=bool)
    bk[L][IdR % Dlast] = True
    if Dlast == 1:
        bk[L][:] = True
    for i in range(L - 1, -1, -1):
        Dl = Ws_d[i].shape[0]
        q, k = np.zeros((Dl, nq), dtype=np.int64), np.zeros(Dl, dtype=bool)
        for (a, b), (s_, t_) in sorted(ents[i].items()):
            if not bk[i + 1][b]:
                continue
            qa = chinfo.make_valid(bq[i + 1][b] - pqs[i][s_] + pqs[i][t_])
            if k[a] and np.any(q[a] != qa):
                raise ValueError("MPO entry (%d,%d) on site %d violates charge conservation" % (a, b, i))
            q[a], k[a] = qa, True
        bq[i], bk[i] = q, k
    bond_q, bond_k = [], []
    for j in range(L + 1):
        both = fk[j] & bk[j]
        if np.any(fq[j][both] != bq[j][both]):
            raise ValueError("MPO charges deduced from the left and from the right disagree on bond %d" % j)
        bond_q.append(np.where(fk[j][:, None], fq[j], bq[j]))
        bond_k.append(fk[j] | bk[j])
    Ws = []
    for i, Wd in enumerate(Ws_d):
        Wd[~bond_k[i]] = 0.
        Wd[:, ~bond_k[i + 1]] = 0.
        p = p_legs[i]
        wL = LegCharge.from_qflat(chinfo, chinfo.make_valid(bond_q[i]), qconj=+1)
        wR = LegCharge.from_qflat(chinfo, chinfo.make_valid(bond_q[i + 1]), qconj=-1)
        W = npc.Array.from_ndarray(Wd, [wL, wR, p, p.conj()], dtype=dtype, qtotal=None if nq == 0 else chinfo.make_valid(),
                                   labels=['wL', 'wR', 'p', 'p*'])
        Ws.append(W)
    return MPO(p_legs, Ws, IdL, IdR)


class MPOEnvironment:
    """``<psi| H |psi>`` environments of a finite chain; ``LP[i]`` is everything left of site i, ``RP[i]`` everything right
    of it."""

    def __init__(self, psi, H):
        self.psi = self.ket = self.bra = psi
        self.H = H
        self.L = psi.L
        self.dtype = np.result_type(psi.dtype, H.dtype)
        self._LP = [None] * self.L
        self._RP = [None] * self.L
        self._heff_cache = {}        # (side, site) -> (env tensor, fused Heff, pipe); see TwoSiteH.combine_Heff
        self._LP[0] = self._boundary(0, left=True)
        self._RP[self.L - 1] = self._boundary(self.L - 1, left=False)

    def _boundary(self, i, left):
        B, W = self.psi.get_B(i, None), self.H.get_W(i)
        leg_ket = B.get_leg('vL' if left else 'vR')
        leg_mpo = W.get_leg('wL' if left else 'wR').conj()
        dense = np.zeros((leg_ket.ind_len, leg_mpo.ind_len, leg_ket.ind_len), dtype=self.dtype)
        idx = (self.H.IdL if left else self.H.IdR) % leg_mpo.ind_len
        dense[np.arange(leg_ket.ind_len), idx, np.arange(leg_ket.ind_len)] = 1.
        if left:
            return npc.Array.from_ndarray(dense, [leg_ket, leg_mpo, leg_ket.conj()], dtype=self.dtype, labels=['vR*', 'wR', 'vR'])
        return npc.Array.from_ndarray(dense, [leg_ket.conj(), leg_mpo, leg_ket], dtype=self.dtype, labels=['vL', 'wL', 'vL*'])

    def get_LP(self, i, store=True):
        """LP left of site i, grown from the nearest stored one."""
        i0 = max(j for j in range(i + 1) if self._LP[j] is not None)
        LP = self._LP[i0]
        for k in range(i0, i):
            LP = self._contract_LP(k, LP)
            if store:
                self._LP[k + 1] = LP
        return LP

    def get_RP(self, i, store=True):
        i0 = min(j for j in range(i, self.L) if self._RP[j] is not None)
        RP = self._RP[i0]
        for k in range(i0, i, -1):
            RP = self._contract_RP(k, RP)
            if store:
                self._RP[k - 1] = RP
        return RP

    def set_LP(self, i, LP, age=None):
        self._LP[i] = LP

    def set_RP(self, i, RP, age=None):
        self._RP[i] = RP

    def invalidate(self, i0, i1, keep_LP, keep_RP):
        """After the tensors of sites i0, i1 changed: drop every stored part that contains one of them, except the one the
        bond update has just computed itself (``LP[i1]`` when moving right, ``RP[i0]`` when moving left)."""
        for j in range(i1 + (1 if keep_LP else 0), self.L):
            if j > 0:
                self._LP[j] = None
        for j in range(i0 - (1 if keep_RP else 0), -1, -1):
            if j < self.L - 1:
                self._RP[j] = None

    def _contract_LP(self, i, LP):
        """LP(i+1) from LP(i): contract with A[i], W[i], A*[i]  (reference mpo.py:3087)."""
        A = self.psi.get_B(i, 'A')
        LP = npc.tensordot(LP, A, axes=('vR', 'vL'))
        LP = npc.tensordot(self.H.get_W(i), LP, axes=(['p*', 'wL'], ['p', 'wR']))
        return npc.tensordot(A.conj(), LP, axes=(['p*', 'vL*'], ['p', 'vR*']))      # 'vR*', 'wR', 'vR'

    def _contract_RP(self, i, RP):
        """RP(i-1) from RP(i): contract with B[i], W[i], B*[i]  (reference mpo.py:3097)."""
        B = self.psi.get_B(i, 'B')
        RP = npc.tensordot(B, RP, axes=('vR', 'vL'))
        RP = npc.tensordot(RP, self.H.get_W(i), axes=(['p', 'wL'], ['p*', 'wR']))
        return npc.tensordot(RP, B.conj(), axes=(['p', 'vL*'], ['p*', 'vR*']))       # 'vL', 'wL', 'vL*'

    def full_contraction(self, i0):
        """<psi|H|psi> evaluated at bond (i0, i0+1)."""
        LP = self.get_LP(i0 + 1, store=False)
        S = self.psi.get_SR(i0)
        RP = self.get_RP(i0, store=False)
        LP = LP.scale_axis(S, 'vR').scale_axis(np.conj(S), 'vR*')
        return npc.inner(LP, RP, axes=(['vR*', 'wR', 'vR'], ['vL*', 'wL', 'vL']), do_conj=False)
