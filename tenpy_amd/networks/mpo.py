"""MPO container and MPO environments -- caller side of the hot path.

Mirrors what a sweep uses from ``tenpy/networks/mpo.py``: ``MPO`` (list of W with labels
``'wL','wR','p','p*'``), ``MPOEnvironment`` with cached ``LP`` / ``RP`` (labels ``'vR*','wR','vR'`` and
``'vL','wL','vL*'``), ``_contract_LP`` / ``_contract_RP`` (reference :3087-3105).  Everything is a
device tensordot; at chi=2048 all L environments (~100 x 2 x 35 MB) stay resident in HBM instead of the
reference's disk cache ("long-context" row).
"""
import numpy as np

from ..linalg import np_conserved as npc
from ..linalg.charges import ChargeInfo, LegCharge

__all__ = ['MPO', 'MPOEnvironment', 'MPOTransferMatrix', 'mpo_from_dense']


class MPO:
    def __init__(self, p_legs, Ws, IdL=0, IdR=-1, bc='finite'):
        self.p_legs = list(p_legs)
        self._W = list(Ws)
        self.L = len(Ws)
        self.IdL = IdL
        self.IdR = IdR
        self.chinfo = Ws[0].chinfo
        self.dtype = Ws[0].dtype
        self.bc = bc
        self.finite = (bc == 'finite')
        self.explicit_plus_hc = False

    def get_W(self, i):
        return self._W[i if self.finite else i % self.L]

    # ---- time evolution operators (Zaletel et al. 2015; reference mpo.py:959-1112, make_W_II :2144) ------------------------
    def make_U(self, dt, approximation='II'):
        if approximation != 'II':
            raise NotImplementedError("tenpy_amd: only the W_II approximation is implemented")
        return self.make_U_II(dt)

    def make_U_II(self, dt):
        """``U_II ~= exp(dt H)`` as an MPO (``dt`` imaginary for real time), IdL = IdR = 0.  The (4d x 4d) exponentials of the
        W_II construction go through the device ``npc.expm`` (charge-less blocks), once per time step size."""
        def expm(h):
            h = np.asarray(h)
            leg = LegCharge.from_trivial(h.shape[0], ChargeInfo())
            return npc.expm(npc.Array.from_ndarray(h, [leg, leg.conj()])).to_ndarray()
        if not self.finite:
            raise NotImplementedError("tenpy_amd: make_U_II for infinite MPO")
        Ws = []
        for i in range(self.L):
            W = self.get_W(i)
            Wd = W.transpose(['wL', 'wR', 'p', 'p*']).to_ndarray()
            DL, DR = Wd.shape[:2]
            IdL_l, IdR_l = self.IdL % DL if DL > 1 else 0, self.IdR % DL if DL > 1 else 0
            IdL_r, IdR_r = self.IdL % DR if DR > 1 else 0, self.IdR % DR if DR > 1 else 0
            if i == 0:
                IdL_l = IdR_l = 0                       # the boundary leg of a finite chain has one entry (= IdL)
            if i == self.L - 1:
                IdL_r = IdR_r = 0                       # ... (= IdR)
            proj_L = np.ones(DL, dtype=bool)
            proj_R = np.ones(DR, dtype=bool)
            if i > 0:
                proj_L[[IdL_l, IdR_l]] = False
            else:
                proj_L[:] = False
            if i < self.L - 1:
                proj_R[[IdL_r, IdR_r]] = False
            else:
                proj_R[:] = False
            D = Wd[IdL_l, IdR_r]
            C = Wd[IdL_l][proj_R]
            B = Wd[proj_L][:, IdR_r]
            A = Wd[proj_L][:, proj_R]
            Ws.append(_make_W_II(dt, A, B, C, D, expm))
        return mpo_from_dense(Ws, self.p_legs, self.chinfo, dtype=np.result_type(dt, self.dtype), IdL=0, IdR=0)

    def apply_naively(self, psi):
        """``psi <- self psi`` without compression: bond dimension chi_MPO * chi (reference :1611).  The Schmidt values
        are only placeholders afterwards; compress or canonicalise."""
        if not (self.finite and psi.finite) or psi.L != self.L:
            raise NotImplementedError("tenpy_amd: apply_naively for finite MPS / MPO of equal length")
        from ..linalg.charges import LegCharge
        L = psi.L

        def plain(leg):        # LegPipe -> LegCharge with the same charge data (reference ``to_LegCharge``)
            return LegCharge.from_qind(leg.chinfo, leg.slices, leg.charges, leg.qconj)
        for i in range(L):
            B = npc.tensordot(psi.get_B(i, 'B'), self.get_W(i), axes=('p', 'p*'))
            if i == 0:
                B = B.take_slice(self.IdL % self.get_W(i).get_leg('wL').ind_len, 'wL')
            if i == L - 1:
                B = B.take_slice(self.IdR % self.get_W(i).get_leg('wR').ind_len, 'wR')
            groups, qc = [], []
            if i > 0:
                groups.append(['wL', 'vL'])
                qc.append(+1)
            if i < L - 1:
                groups.append(['wR', 'vR'])
                qc.append(-1)
            if groups:
                B = B.combine_legs(groups, qconj=qc)
                B.ireplace_labels(['(wL.vL)', '(wR.vR)'][(0 if i > 0 else 1):(2 if i < L - 1 else 1)],
                                  ['vL', 'vR'][(0 if i > 0 else 1):(2 if i < L - 1 else 1)])
                for lab in ('vL', 'vR'):
                    a = B.get_leg_index(lab)
                    if hasattr(B.legs[a], 'q_map'):
                        B.legs[a] = plain(B.legs[a])
                        B._skey = None
            psi.set_B(i, B, 'B')
        psi.set_SL(0, np.ones(psi.get_B(0, None).get_leg('vL').ind_len))
        for i in range(L):
            psi.set_SR(i, np.ones(psi.get_B(i, None).get_leg('vR').ind_len))

    def apply(self, psi, options):
        """Apply to ``psi`` in place and compress; ``options['compression_method']`` = 'SVD' (reference :1562)."""
        method = options.get('compression_method', 'SVD')
        if method != 'SVD':
            raise NotImplementedError("tenpy_amd: compression_method %r" % (method,))
        self.apply_naively(psi)
        return psi.compress_svd(dict(options.get('trunc_params', {})))

    @property
    def chi(self):
        return [W.get_leg('wR').ind_len for W in self._W[:-1]]


def _make_W_II(t, A, B, C, D, expm):
    """W_II tensor (Zaletel et al. 2015, Eq. 11) from the blocks of ``W = [[1, C, D], [0, A, B], [0, 0, 1]]``: for every
    (row, column) of A one exponential in a space extended by two hard-core bosons."""
    tC = np.sqrt(np.abs(t))
    tB = t / tC
    d = D.shape[0]
    Nr, Nc = A.shape[0], A.shape[1]
    W = np.zeros((1 + Nr, 1 + Nc, d, d), dtype=np.result_type(D, t))
    Id2 = np.eye(2)
    b = np.array([[0., 0.], [1., 0.]])
    Id4, Br, Bc, Brc = np.kron(Id2, Id2), np.kron(b, Id2), np.kron(Id2, b), np.kron(b, b)

    def part(h):
        return expm(h).reshape((2, 2, d, 2, 2, d))[:, :, :, 0, 0, :]
    for r in range(Nr):
        for c in range(Nc):
            w = part(np.kron(Brc, A[r, c]) + np.kron(Br, tB * B[r]) + np.kron(Bc, tC * C[c]) + t * np.kron(Id4, D))
            W[1 + r, 1 + c] = w[1, 1]
            if c == 0:
                W[1 + r, 0] = w[1, 0]
            if r == 0:
                W[0, 1 + c] = w[0, 1]
                if c == 0:
                    W[0, 0] = w[0, 0]
        if Nc == 0:
            w = part(np.kron(Br, tB * B[r]) + t * np.kron(Id4, D))
            W[1 + r, 0] = w[1, 0]
            if r == 0:
                W[0, 0] = w[0, 0]
    if Nr == 0:
        for c in range(Nc):
            w = part(np.kron(Bc, tC * C[c]) + t * np.kron(Id4, D))
            W[0, 1 + c] = w[0, 1]
            if c == 0:
                W[0, 0] = w[0, 0]
        if Nc == 0:
            W = expm(t * D).reshape([1, 1, d, d])
    return W


def mpo_from_dense(W_dense_list, p_legs, chinfo, dtype=np.float64, IdL=0, IdR=-1, bc='finite'):
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
    if bc == 'infinite':
        # the bond right of the last site IS the bond left of the first one: let both passes go around until nothing changes
        for _ in range(2 * Ws_d[0].shape[0] + 2):
            changed = False
            new_k = fk[0] | fk[L]
            if np.any(new_k != fk[0]) or np.any(new_k != fk[L]):
                q = np.where(fk[0][:, None], fq[0], fq[L])
                fq[0], fk[0] = q, new_k
                changed = True
                for i in range(L):
                    for (a, b), (s_, t_) in sorted(ents[i].items()):
                        if fk[i][a] and not fk[i + 1][b]:
                            fq[i + 1][b] = chinfo.make_valid(fq[i][a] + pqs[i][s_] - pqs[i][t_])
                            fk[i + 1][b] = True
            new_k = bk[0] | bk[L]
            if np.any(new_k != bk[0]) or np.any(new_k != bk[L]):
                q = np.where(bk[L][:, None], bq[L], bq[0])
                bq[L], bk[L] = q, new_k
                changed = True
                for i in range(L - 1, -1, -1):
                    for (a, b), (s_, t_) in sorted(ents[i].items()):
                        if bk[i + 1][b] and not bk[i][a]:
                            bq[i][a] = chinfo.make_valid(bq[i + 1][b] - pqs[i][s_] + pqs[i][t_])
                            bk[i][a] = True
            if not changed:
                break
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
    if bc == 'infinite':
        Ws[0].get_leg('wL').test_contractible(Ws[-1].get_leg('wR'))
    return MPO(p_legs, Ws, IdL, IdR, bc=bc)


class MPOEnvironment:
    """``<bra| H |ket>`` environments with bra = ket = psi; LP[i] is everything left of site i.  For an infinite MPS the
    indices are taken modulo L: the stored LP[i] is the one most recently computed for ANY of the equivalent sites
    i + n L (reference ``BaseEnvironment.get_LP``, mps.py:6429), and every part carries an ``age`` = number of physical
    sites it contains (used for the energy per site of iDMRG)."""

    def __init__(self, psi, H):
        self.psi = self.ket = self.bra = psi
        self.H = H
        self.L = psi.L
        self.finite = psi.finite
        self.dtype = np.result_type(psi.dtype, H.dtype)
        self._LP = [None] * self.L
        self._heff_cache = {}        # (side, site) -> (env tensor, fused Heff, pipe); see TwoSiteH.combine_Heff
        self._RP = [None] * self.L
        self._LP_age = [None] * self.L
        self._RP_age = [None] * self.L
        init_LP = init_RP = None
        if not self.finite:
            # reference init_first_LP_last_RP (mpo.py:2806-2849): for an infinite MPS in canonical form start from the
            # dominant (generalised) eigenvectors of the MPO transfer matrix of the unit cell
            if float(np.linalg.norm(psi.norm_error())) > 1.e-10:
                psi.canonical_form()
            init_RP = MPOTransferMatrix(H, psi, transpose=False).dominant_eigenvector()[1]
            init_LP = MPOTransferMatrix(H, psi, transpose=True).dominant_eigenvector()[1]
        self.set_LP(0, init_LP if init_LP is not None else self.init_LP(0), age=0)
        self.set_RP(self.L - 1, init_RP if init_RP is not None else self.init_RP(self.L - 1), age=0)

    def _idx(self, i):
        if self.finite:
            if not 0 <= i < self.L:
                raise IndexError("environment index %d out of range" % i)
            return i
        return i % self.L

    def init_LP(self, i):
        leg_ket = self.psi.get_B(i, None).get_leg('vL')
        leg_mpo = self.H.get_W(i).get_leg('wL').conj()
        dense = np.zeros((leg_ket.ind_len, leg_mpo.ind_len, leg_ket.ind_len), dtype=self.dtype)
        IdL = self.H.IdL % leg_mpo.ind_len
        for j in range(leg_ket.ind_len):
            dense[j, IdL, j] = 1.
        return npc.Array.from_ndarray(dense, [leg_ket, leg_mpo, leg_ket.conj()], dtype=self.dtype, labels=['vR*', 'wR', 'vR'])

    def init_RP(self, i):
        leg_ket = self.psi.get_B(i, None).get_leg('vR')
        leg_mpo = self.H.get_W(i).get_leg('wR').conj()
        dense = np.zeros((leg_ket.ind_len, leg_mpo.ind_len, leg_ket.ind_len), dtype=self.dtype)
        IdR = self.H.IdR % leg_mpo.ind_len
        for j in range(leg_ket.ind_len):
            dense[j, IdR, j] = 1.
        return npc.Array.from_ndarray(dense, [leg_ket.conj(), leg_mpo, leg_ket], dtype=self.dtype,
                                      labels=['vL', 'wL', 'vL*'])

    # The environment legs: LP has ('vR*', 'wR', 'vR') where 'vR' contracts with the ket's 'vL'.
    def get_LP(self, i, store=True):
        """LP left of site i from the nearest stored one (at most L sites to the left)."""
        for i0 in range(i, i - self.L, -1):
            if not self.finite or i0 >= 0:
                LP = self._LP[self._idx(i0)]
                if LP is not None:
                    break
        else:
            raise ValueError("No left part in the system???")
        age = self._LP_age[self._idx(i0)] or 0
        for k in range(i0, i):
            LP = self._contract_LP(k, LP)
            age = age + 1
            if store:
                self.set_LP(k + 1, LP, age=age)
        return LP

    def get_RP(self, i, store=True):
        for i0 in range(i, i + self.L):
            if not self.finite or i0 < self.L:
                RP = self._RP[self._idx(i0)]
                if RP is not None:
                    break
        else:
            raise ValueError("No right part in the system???")
        age = self._RP_age[self._idx(i0)] or 0
        for k in range(i0, i, -1):
            RP = self._contract_RP(k, RP)
            age = age + 1
            if store:
                self.set_RP(k - 1, RP, age=age)
        return RP

    def get_LP_age(self, i):
        return self._LP_age[self._idx(i)]

    def get_RP_age(self, i):
        return self._RP_age[self._idx(i)]

    def set_LP(self, i, LP, age=None):
        i = self._idx(i)
        self._LP[i] = LP
        if age is not None:
            self._LP_age[i] = age

    def set_RP(self, i, RP, age=None):
        i = self._idx(i)
        self._RP[i] = RP
        if age is not None:
            self._RP_age[i] = age

    def del_LP(self, i):
        self._LP[self._idx(i)] = None

    def del_RP(self, i):
        self._RP[self._idx(i)] = None

    def _contract_LP(self, i, LP):
        """LP(i+1) from LP(i): contract with A[i], W[i], A*[i]  (reference mpo.py:3087)."""
        A = self.psi.get_B(i, 'A')
        LP = npc.tensordot(LP, A, axes=('vR', 'vL'))
        LP = npc.tensordot(self.H.get_W(i), LP, axes=(['p*', 'wL'], ['p', 'wR']))
        LP = npc.tensordot(A.conj(), LP, axes=(['p*', 'vL*'], ['p', 'vR*']))
        return LP      # labels 'vR*', 'wR', 'vR'

    def _contract_RP(self, i, RP):
        """RP(i-1) from RP(i): contract with B[i], W[i], B*[i]  (reference mpo.py:3097)."""
        B = self.psi.get_B(i, 'B')
        RP = npc.tensordot(B, RP, axes=('vR', 'vL'))
        RP = npc.tensordot(RP, self.H.get_W(i), axes=(['p', 'wL'], ['p*', 'wR']))
        RP = npc.tensordot(RP, B.conj(), axes=(['p', 'vL*'], ['p*', 'vR*']))
        return RP      # labels 'vL', 'wL', 'vL*'

    def full_contraction(self, i0):
        """<psi|H|psi> evaluated at bond (i0, i0+1)."""
        LP = self.get_LP(i0 + 1, store=False)
        S = self.psi.get_SR(i0)
        RP = self.get_RP(i0, store=False)
        if isinstance(S, npc.Array):     # general bond matrix of a sweep with mixer (reference mps.py:6715-6724)
            LP = npc.tensordot(S.conj(), LP, axes=['vL*', 'vR*'])
            LP = npc.tensordot(LP, S, axes=['vR', 'vL'])
        else:
            LP = LP.scale_axis(S, 'vR').scale_axis(np.conj(S), 'vR*')
        return npc.inner(LP, RP, axes=(['vR*', 'wR', 'vR'], ['vL*', 'wL', 'vL']), do_conj=False)


class MPOTransferMatrix:
    """Transfer matrix of ``<psi| H |psi>`` for one unit cell of an infinite MPS in canonical form (reference mpo.py:3694).

    Its Jordan block (the energy grows by ``e L`` per application) is removed by projecting the identity component out
    after every application (``_project``), which leaves a translation invariant fixed point: the environment an infinite
    DMRG / TDVP run starts from.  ``transpose=False``: right environments, vectors [vL, wL, vL*]; ``transpose=True``:
    left environments, vectors [vR*, wR, vR].  The dominant eigenvector is found by the device ``Arnoldi``."""

    def __init__(self, H, psi, transpose=False):
        if psi.finite or H.finite:
            raise ValueError("Only makes sense for infinite MPS")
        if H.L != psi.L:
            raise NotImplementedError("tenpy_amd: MPO and MPS unit cells must have the same length")
        self.L = L = psi.L
        self.dtype = np.result_type(psi.dtype, H.dtype)
        self.transpose = transpose
        IdL, IdR = H.IdL, H.IdR
        S = psi.get_SL(0)
        if isinstance(S, npc.Array):
            raise NotImplementedError("tenpy_amd: MPOTransferMatrix needs diagonal Schmidt values")
        if not transpose:
            wR = H.get_W(L - 1).get_leg('wR')
            wL = wR.conj()
            vR = psi.get_B(L - 1, 'B').get_leg('vR')
            rho = npc.diag(S**2, vR, dtype=self.dtype, labels=['vR', 'vR*'])
            self.acts_on = ['vL', 'wL', 'vL*']
            self._M = [psi.get_B(i, 'B').astype(self.dtype, copy=False) for i in reversed(range(L))]
            self._W = [H.get_W(i).astype(self.dtype, copy=False) for i in reversed(range(L))]
            self._chi0 = vR.ind_len
            eye = npc.diag(1., vR.conj(), dtype=self.dtype, labels=['vL', 'vL*'])
            self._E_shift = eye.add_leg(wL, IdL % wL.ind_len, axis=1, label='wL')
            self._proj_norm = eye.add_leg(wL, IdR % wL.ind_len, axis=1, label='wL').conj()      # vL* wL* vL
            self._proj_rho = rho.add_leg(wR, IdL % wR.ind_len, axis=1, label='wR')             # vR wR vR*
            self.guess = eye.add_leg(wL, IdR % wL.ind_len, axis=1, label='wL')
        else:
            wL = H.get_W(0).get_leg('wL')
            wR = wL.conj()
            vL = psi.get_B(0, 'A').get_leg('vL')
            rho = npc.diag(S**2, vL.conj(), dtype=self.dtype, labels=['vL*', 'vL'])
            self.acts_on = ['vR*', 'wR', 'vR']
            self._M = [psi.get_B(i, 'A').astype(self.dtype, copy=False) for i in range(L)]
            self._W = [H.get_W(i).astype(self.dtype, copy=False) for i in range(L)]
            self._chi0 = vL.ind_len
            eye = npc.diag(1., vL, dtype=self.dtype, labels=['vR*', 'vR'])
            self._E_shift = eye.add_leg(wR, IdR % wR.ind_len, axis=1, label='wR')
            self._proj_norm = eye.add_leg(wR, IdL % wR.ind_len, axis=1, label='wR').conj()      # vR wR* vR*
            self._proj_rho = rho.add_leg(wL, IdR % wL.ind_len, axis=1, label='wL')             # vL* wL vL
            self.guess = eye.add_leg(wR, IdL % wR.ind_len, axis=1, label='wR')
        self._M_conj = [M.conj() for M in self._M]

    def matvec(self, vec, project=True):
        if not self.transpose:
            for Bc, W, B in zip(self._M_conj, self._W, self._M):
                vec = npc.tensordot(B, vec, axes=['vR', 'vL'])                              # vL p wL vL*
                vec = npc.tensordot(vec, W, axes=[['p', 'wL'], ['p*', 'wR']])               # vL vL* wL p
                vec = npc.tensordot(vec, Bc, axes=[['vL*', 'p'], ['vR*', 'p*']])            # vL wL vL*
        else:
            for Ac, W, A in zip(self._M_conj, self._W, self._M):
                vec = npc.tensordot(vec, A, axes=['vR', 'vL'])                              # vR* wR p vR
                vec = npc.tensordot(W, vec, axes=[['wL', 'p*'], ['wR', 'p']])               # wR p vR* vR
                vec = npc.tensordot(Ac, vec, axes=[['p*', 'vL*'], ['p', 'vR*']])            # vR* wR vR
        if list(vec.get_leg_labels()) != self.acts_on:
            vec = vec.transpose(self.acts_on)
        return self._project(vec) if project else vec

    def _project(self, vec):
        """Remove the additive energy part (``T RP = RP + e 1``) measured against the density matrix ('rho' gauge)."""
        axes = (['vL', 'wL', 'vL*'], ['vR', 'wR', 'vR*']) if not self.transpose else (['vR*', 'wR', 'vR'], ['vL*', 'wL', 'vL'])
        E = npc.inner(vec, self._proj_rho, axes=axes, do_conj=False)
        res = vec.copy(deep=True)
        res.iadd_prefactor_other(-E, self._E_shift)
        return res

    def dominant_eigenvector(self, **arnoldi_params):
        """Returns ``(eigenvalue ~ 1, environment)`` normalised such that its identity component is 1."""
        from ..linalg.krylov_based import Arnoldi
        opts = dict(N_min=2, N_max=40, P_tol=1.e-28, which='LM')
        opts.update(arnoldi_params)
        vec = self.guess
        val = None
        for _ in range(20):                 # restarts: the Krylov space is short, the gap of a product state is not
            vals, vecs, N = Arnoldi(self, vec, opts).run()
            vec, val = vecs[0], vals[0]
            if N < opts['N_max']:
                break
        nrm = npc.inner(self._proj_norm, vec, axes='range', do_conj=False) / self._chi0
        return val, vec * (1. / nrm)

    def energy(self, dom_vec):
        """Energy per site from the growth of the un-projected application (reference :3911)."""
        axes = (['vL', 'wL', 'vL*'], ['vR', 'wR', 'vR*']) if not self.transpose else (['vR*', 'wR', 'vR'], ['vL*', 'wL', 'vL'])
        E0 = npc.inner(dom_vec, self._proj_rho, axes=axes, do_conj=False)
        E = npc.inner(self.matvec(dom_vec, project=False), self._proj_rho, axes=axes, do_conj=False)
        return (E - E0) / self.L
