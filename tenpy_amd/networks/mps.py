"""MPS holding device-resident site tensors -- the *caller side* of the hot path.

Not a re-implementation of ``tenpy/networks/mps.py`` (7.6k lines): it holds what the DMRG / TEBD / TDVP engines
of this package touch -- ``from_product_state``, ``get_B`` / ``set_B`` / ``get_theta`` / ``set_SL`` / ``set_SR`` with the reference's leg
labels ``('vL', 'p', 'vR')`` and canonical-form convention ``B = S**nuL  Gamma  S**nuR`` ('A' = (1,0), 'B' = (0,1), 'Th' = (1,1)),
finite and infinite boundary conditions (indices modulo the unit cell), 2-D bond matrices of a mixer sweep (incl. their
pseudo-inverse), ``canonical_form`` (finite), ``expectation_value``, ``overlap``, ``entanglement_entropy`` -- and
``MPSEnvironment`` for overlaps with other states.
"""
import numpy as np

from ..linalg import np_conserved as npc
from ..linalg.charges import LegCharge

__all__ = ['MPS', 'MPSEnvironment', 'TransferMatrix']

_FORMS = {'A': (1., 0.), 'B': (0., 1.), 'C': (0.5, 0.5), 'G': (0., 0.), 'Th': (1., 1.), None: None}


class MPS:
    def __init__(self, p_legs, Bs, SVs, form='B', bc='finite'):
        """``bc='infinite'``: the L tensors are the unit cell of an infinite MPS; site and bond indices are taken modulo L
        (``_S[L]`` is kept identical to ``_S[0]``), as in the reference (mps.py ``_to_valid_site_index``)."""
        if bc not in ('finite', 'infinite'):
            raise ValueError("bc must be 'finite' or 'infinite'")
        self.p_legs = list(p_legs)          # physical leg of each site
        self.L = len(Bs)
        self._B = list(Bs)
        self._S = [np.asarray(s, dtype=np.float64) for s in SVs]   # L+1 Schmidt spectra, host
        self.form = [_FORMS[form]] * self.L
        self.chinfo = Bs[0].chinfo
        self.dtype = Bs[0].dtype
        self.bc = bc
        self.finite = (bc == 'finite')
        self.norm = 1.                       # tracked by the time-evolution engines (reference MPS.norm)

    @classmethod
    def from_product_state(cls, p_legs, p_state, dtype=np.float64, bc='finite'):
        """Product state; ``p_state[i]`` is the flat physical index occupied on site i.  For ``bc='infinite'`` the unit
        cell must be charge neutral (no charge shift between unit cells is implemented)."""
        chinfo = p_legs[0].chinfo
        L = len(p_legs)
        Bs = []
        q_left = chinfo.make_valid()
        for i in range(L):
            leg_p = p_legs[i]
            qi, _ = leg_p.get_qindex(int(p_state[i]))
            q_right = chinfo.make_valid(q_left + leg_p.get_charge(qi))
            vL = LegCharge.from_qflat(chinfo, [q_left], qconj=+1)
            vR = LegCharge.from_qflat(chinfo, [q_right], qconj=-1)
            dense = np.zeros((1, leg_p.ind_len, 1), dtype=dtype)
            dense[0, int(p_state[i]), 0] = 1.
            Bs.append(npc.Array.from_ndarray(dense, [vL, leg_p, vR], dtype=dtype, labels=['vL', 'p', 'vR']))
            q_left = q_right
        if bc == 'infinite' and np.any(q_left != chinfo.make_valid()):
            raise ValueError("infinite MPS: the unit cell of the product state must have total charge 0")
        return cls(p_legs, Bs, [np.ones(1)] * (L + 1), form='B', bc=bc)

    @property
    def chi(self):
        """Bond dimensions (finite: the L-1 inner bonds; infinite: the bond left of every site); a 2-D bond matrix (DMRG
        with mixer) counts with its smaller dimension (reference mps.py)."""
        Ss = self._S[1:-1] if self.finite else self._S[:self.L]      # (reference: MPS.nontrivial_bonds)
        return [int(min(s.shape)) if isinstance(s, npc.Array) else len(s) for s in Ss]

    def _site(self, i):
        if self.finite:
            if not 0 <= i < self.L:
                raise IndexError("site index %d out of range for a finite MPS of length %d" % (i, self.L))
            return i
        return i % self.L

    def _bond(self, i):
        """Index into ``_S`` of the bond LEFT of site i."""
        if self.finite:
            return i
        return i % self.L

    def get_SL(self, i):
        return self._S[self._bond(i)]

    def get_SR(self, i):
        return self._S[self._bond(i + 1)] if not self.finite else self._S[i + 1]

    def set_SL(self, i, S):
        """Schmidt values (1-D host array) or, during DMRG with a mixer, a general bond MATRIX (2-D device Array with
        labels 'vL', 'vR') left of site i  (reference ``MPS.set_SL``; 2-D case: mps.py:5970)."""
        S = S if isinstance(S, npc.Array) else np.asarray(S)
        b = self._bond(i)
        self._S[b] = S
        if not self.finite and b == 0:
            self._S[self.L] = S

    def set_SR(self, i, S):
        if self.finite:
            self._S[i + 1] = S if isinstance(S, npc.Array) else np.asarray(S)
        else:
            self.set_SL(i + 1, S)

    def set_B(self, i, B, form='B'):
        i = self._site(i)
        self._B[i] = B.transpose(['vL', 'p', 'vR']) if B._labels != ['vL', 'p', 'vR'] else B
        self.form[i] = _FORMS[form] if not isinstance(form, tuple) else form

    @staticmethod
    def _scale_axis_B(B, S, power, axis, cutoff=1.e-16):
        """``B.scale_axis(S**power)``; a 2-D bond matrix (DMRG with a mixer) is contracted instead, its Moore-Penrose
        pseudo-inverse (``npc.pinv`` with ``cutoff``) for power -1  (reference mps.py:5964-6002)."""
        if power == 0.:
            return B
        if isinstance(S, npc.Array):
            if power == -1.:
                S = npc.pinv(S, cutoff).iset_leg_labels(['vL', 'vR'])
            elif power != 1.:
                raise ValueError("Can't scale/tensordot a 2D `S` with power %r" % (power,))
            labels = B.get_leg_labels()
            if axis == 'vL':
                B = npc.tensordot(S, B, axes=['vR', 'vL'])
            else:
                B = npc.tensordot(B, S, axes=['vR', 'vL'])
            return B.transpose(labels)
        if power == 1.:
            return B.scale_axis(S, axis)
        return B.scale_axis(S**power, axis)

    def get_B(self, i, form='B', copy=False, cutoff=1.e-16):
        """Site tensor converted to ``form``; ``form=None`` returns the stored tensor."""
        want = _FORMS[form] if not isinstance(form, tuple) else form
        i = self._site(i)
        B = self._B[i]
        if want is not None and want != self.form[i]:
            have = self.form[i]
            B = self._scale_axis_B(B, self._S[i], want[0] - have[0], 'vL', cutoff)
            B = self._scale_axis_B(B, self._S[i + 1], want[1] - have[1], 'vR', cutoff)
        elif copy:
            B = B.copy(deep=True)
        return B

    def get_theta(self, i, n=2, formL=1., formR=1., cutoff=1.e-16):
        """Two-site wave function with labels ``'vL', 'p0', 'p1', 'vR'``, or the one-site one ``'vL', 'p0', 'vR'``
        (reference mps.py:3041; n=1: ``get_B(i, (1., 1.))`` :3075)."""
        if n == 1:
            return self.get_B(i, (formL, formR), cutoff=cutoff).replace_label('p', 'p0')
        assert n == 2
        i, i1 = self._site(i), self._site(i + 1)
        B0 = self._B[i]
        B0 = self._scale_axis_B(B0, self._S[i], formL - self.form[i][0], 'vL', cutoff)
        B0 = self._scale_axis_B(B0, self._S[i + 1], 1. - self.form[i][1] - self.form[i1][0], 'vR', cutoff)
        B1 = self._B[i1]
        B1 = self._scale_axis_B(B1, self._S[i1 + 1], formR - self.form[i1][1], 'vR', cutoff)
        B0 = B0.replace_label('p', 'p0')
        B1 = B1.replace_label('p', 'p1')
        return npc.tensordot(B0, B1, axes=['vR', 'vL'])

    def canonical_form(self, renormalize=True, cutoff=0.):
        """Bring a finite MPS into canonical (B) form with correct Schmidt values (reference ``canonical_form_finite``,
        mps.py:4505-4603): a left-to-right sweep of block QR decompositions, then a right-to-left sweep of block SVDs.
        With ``renormalize=False`` the norm of the state is multiplied into ``self.norm``."""
        if not self.finite:
            return self._canonical_form_infinite(renormalize=renormalize, cutoff=cutoff if cutoff else 1.e-15)
        L = self.L
        self.set_SL(0, np.array([1.]))
        self.set_SR(L - 1, np.array([1.]))

        def normalise(M):
            nrm = float(npc.norm(M))
            if not renormalize:
                self.norm = self.norm * nrm
            M.iscale_prefactor(1. / nrm)
            return M
        if any(f is None for f in self.form):
            M, form = self.get_B(0, None), None           # no canonical form before: ignore the stored S
        else:
            M, form = self.get_B(0, 'Th'), 'B'
        M = normalise(M.copy(deep=True))
        Q, R = npc.qr(M.combine_legs(['vL', 'p']), inner_labels=['vR', 'vL'])
        self.set_B(0, Q.split_legs(0), form='A')
        for i in range(1, L):
            M = npc.tensordot(R, self.get_B(i, form), axes=['vR', 'vL'])
            M = normalise(M)
            if i == L - 1:
                break
            Q, R = npc.qr(M.combine_legs(['vL', 'p']), inner_labels=['vR', 'vL'])
            self.set_B(i, Q.split_legs(0), form='A')
        kw = dict(inner_labels=['vR', 'vL'])
        if cutoff:
            kw['cutoff'] = cutoff
        U, S, V = npc.svd(M.combine_legs(['p', 'vR'], qconj=-1), **kw)
        if not renormalize:
            self.norm = self.norm * float(np.linalg.norm(S))
        S = S / np.linalg.norm(S)
        self.set_SL(L - 1, S)
        self.set_B(L - 1, V.split_legs(1), form='B')
        for i in range(L - 2, -1, -1):
            M = npc.tensordot(self.get_B(i, 'A'), U.scale_axis(S, 'vR'), axes=['vR', 'vL'])
            U, S, V = npc.svd(M.combine_legs(['p', 'vR'], qconj=-1), qtotal_LR=[None, M.qtotal], **kw)
            S = S / np.linalg.norm(S)
            self.set_SL(i, S)
            self.set_B(i, V.split_legs(1), form='B')
        assert len(S) == 1
        self._B[0] = self._B[0] * U.to_ndarray()[0, 0]       # a trivial phase factor, but better keep it

    # ---- infinite MPS: Algorithms 1, 2 of Vanderstraeten, Haegeman, Verstraete 2019 (reference canonical_form_infinite2, :4721) ----
    def _canonical_form_infinite(self, renormalize=True, tol=1.e-15, arnoldi_params=None, cutoff=1.e-15):
        from ..linalg.krylov_based import Arnoldi
        assert cutoff <= tol or True
        L = self.L
        ap = dict(arnoldi_params or {})
        if any(f is None for f in self.form):
            self.form = [_FORMS['B']] * L
            self._S[0] = self._S[L] = np.ones(self._B[0].get_leg('vL').ind_len)
        else:
            for i in range(L):
                self._B[i] = self.get_B(i, 'B')
            self.form = [_FORMS['B']] * L

        def qr_R2L(R):         # B[0] ... B[L-1] R  ->  R Q[0] ... Q[L-1]
            Qs = [None] * L
            for i in reversed(range(L)):
                BR = npc.tensordot(self._B[i], R, axes=['vR', 'vL']).combine_legs(['p', 'vR'], new_axes=0, qconj=-1)
                Q, R = npc.qr(BR, inner_labels=['vL', 'vR'], pos_diag_R=True, qtotal_Q=BR.qtotal, inner_qconj=-1)
                Qs[i] = Q.split_legs()
            return Qs, R

        def qr_L2R(Lm):        # L B[0] ... B[L-1]  ->  Q[0] ... Q[L-1] L
            Qs = [None] * L
            for i in range(L):
                LB = npc.tensordot(Lm, self._B[i], axes=['vR', 'vL']).combine_legs(['vL', 'p'], new_axes=0, qconj=+1)
                Q, Lm = npc.qr(LB, inner_labels=['vR', 'vL'], pos_diag_R=True, qtotal_Q=LB.qtotal, inner_qconj=+1)
                Qs[i] = Q.split_legs()
            return Qs, Lm

        def fixed_point(M, sweep, right):
            for _ in range(10000):
                M = M / npc.norm(M)
                old = M
                new_Bs, M = sweep(M)
                nrm = npc.norm(M)
                M = M / nrm
                M = M.transpose(old.get_leg_labels())
                err = npc.norm(M - old)
                if err <= tol:
                    return new_Bs, M, nrm
                ap['E_tol'] = err / 10.
                TM = TransferMatrix(new_Bs, self._B, transpose=not right)
                vec = M.replace_label('vR', 'vL*') if right else M.replace_label('vL', 'vR*')
                _, vecs, _ = Arnoldi(TM, vec, ap).run()
                M = vecs[0].replace_label('vL*', 'vR') if right else vecs[0].replace_label('vR*', 'vL')
                if right:
                    _, M = npc.qr(M.transpose(['vR', 'vL']), inner_labels=['vL', 'vR'], pos_diag_R=True, inner_qconj=-1)
                else:
                    _, M = npc.qr(M.transpose(['vL', 'vR']), inner_labels=['vR', 'vL'], pos_diag_R=True, inner_qconj=+1)
            raise RuntimeError("canonical_form did not converge up to tol=%g (last error %g)" % (tol, err))
        R_guess = npc.diag(1., self._B[0].get_leg('vL'), labels=['vL', 'vR'])
        new_Bs, _, nrm = fixed_point(R_guess, qr_R2L, True)
        if not renormalize:
            self.norm *= nrm
        self._B = new_Bs
        C_guess = npc.diag(self.get_SL(0), self._B[0].get_leg('vL'), labels=['vL', 'vR'])
        new_As, C, _ = fixed_point(C_guess, qr_L2R, False)
        C = C.transpose(['vL', 'vR'])
        U, S, V = npc.svd(C, cutoff=cutoff, inner_labels=['vR', 'vL'])
        new_As[0] = npc.tensordot(U.conj().ireplace_label('vR*', 'vL'), new_As[0], axes=['vL*', 'vL'])
        for i in reversed(range(L)):
            th = npc.tensordot(new_As[i], U.scale_axis(S, 'vR'), axes=['vR', 'vL'])
            th = th.combine_legs(['p', 'vR'], new_axes=1)
            U, S, V = npc.svd(th, cutoff=cutoff, inner_labels=['vR', 'vL'])
            self._B[i] = V.split_legs().transpose(['vL', 'p', 'vR'])
            self.set_SL(i, S)
        self._B[L - 1] = npc.tensordot(self._B[L - 1], U, axes=['vR', 'vL'])

    def norm_error(self):
        """What the reference calls ``MPS.norm_test()`` (mps.py:4432): for every site the deviations
        ``| theta theta^dagger - S_L^2 |`` and ``| theta^dagger theta - S_R^2 |`` of the reduced density matrices from the
        stored Schmidt values, shape (L, 2); zero in canonical form."""
        err = np.empty((self.L, 2))
        for i in range(self.L):
            th = self.get_B(i, 'Th')
            for k, (ax, ax_c, lab, S) in enumerate(((['p', 'vR'], ['p*', 'vR*'], 'vL', self.get_SL(i)),
                                                    (['vL', 'p'], ['vL*', 'p*'], 'vR', self.get_SR(i)))):
                rho = npc.tensordot(th, th.conj(), axes=[ax, ax_c])
                if isinstance(S, npc.Array):
                    rho2 = npc.tensordot(S, S.conj(), axes=(['vR', 'vR*'] if k == 0 else ['vL', 'vL*']))
                else:
                    rho2 = npc.diag(S**2, rho.legs[0], dtype=rho.dtype)
                err[i, k] = npc.norm(rho - rho2.iset_leg_labels(rho.get_leg_labels()))
        return err

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

    def correlation_function(self, op1, op2, sites1=None, sites2=None, opstr=None):
        """``C[i, j] = <psi| op1_i op2_j |psi>`` for one-site operators given as dense (d, d) host matrices [p, p*] or device
        Arrays with labels 'p', 'p*' (reference ``MPS.correlation_function`` for ``str_on_first=True`` semantics: the operator
        string ``opstr`` -- e.g. the Jordan-Wigner sign -- acts on the sites strictly between i and j and, multiplied onto
        ``op1``, on site min(i, j) when given).  Finite MPS; every entry is a chain of device tensordots."""
        if not self.finite:
            raise NotImplementedError("tenpy_amd: correlation_function of infinite MPS")
        L = self.L
        sites1 = list(range(L)) if sites1 is None else list(sites1)
        sites2 = list(range(L)) if sites2 is None else list(sites2)

        def as_op(op, i):
            if isinstance(op, npc.Array):
                return op
            leg = self._B[i].get_leg('p')
            return npc.Array.from_ndarray(np.asarray(op), [leg, leg.conj()], labels=['p', 'p*'], cutoff=0.)
        res = np.zeros((len(sites1), len(sites2)), dtype=np.complex128)
        for a, i in enumerate(sites1):
            for b, j in enumerate(sites2):
                if i == j:
                    O = npc.tensordot(as_op(op1, i), as_op(op2, i), axes=['p*', 'p'])
                    th = self.get_B(i, 'Th')
                    res[a, b] = npc.inner(th, npc.tensordot(O, th, axes=['p*', 'p']), axes='labels', do_conj=True)
                    continue
                lo, hi = (i, j) if i < j else (j, i)
                O_lo, O_hi = (as_op(op1, i), as_op(op2, j)) if i < j else (as_op(op2, j), as_op(op1, i))
                if opstr is not None:
                    S_lo = as_op(opstr, lo)
                    O_lo = npc.tensordot(O_lo, S_lo, axes=['p*', 'p']) if i < j else npc.tensordot(S_lo, O_lo, axes=['p*', 'p'])
                th = self.get_B(lo, 'Th')            # orthogonality centre on the left site: everything to its left drops out
                C = npc.tensordot(O_lo, th, axes=['p*', 'p'])
                C = npc.tensordot(th.conj(), C, axes=[['vL*', 'p*'], ['vL', 'p']])           # vR*, vR
                for k in range(lo + 1, hi):
                    B = self.get_B(k, 'B')
                    C = npc.tensordot(C, B, axes=['vR', 'vL'])
                    if opstr is not None:
                        C = npc.tensordot(as_op(opstr, k), C, axes=['p*', 'p'])
                    C = npc.tensordot(B.conj(), C, axes=[['vL*', 'p*'], ['vR*', 'p']])
                B = self.get_B(hi, 'B')
                C = npc.tensordot(C, B, axes=['vR', 'vL'])
                C = npc.tensordot(O_hi, C, axes=['p*', 'p'])
                res[a, b] = npc.inner(B.conj(), C, axes=[['vL*', 'p*', 'vR*'], ['vR*', 'p', 'vR']], do_conj=False)
        return np.real_if_close(res)

    def overlap(self, other):
        """``<self|other>`` including the norms of both states (reference ``MPS.overlap`` for finite MPS)."""
        if self.finite != other.finite:
            raise ValueError("can't take overlap between MPS with different bc")
        if not self.finite:
            # per unit cell: the dominant eigenvalue of the mixed transfer matrix in the zero-charge sector (reference :4278)
            from ..linalg.krylov_based import Arnoldi
            if self.L != other.L:
                raise NotImplementedError("tenpy_amd: overlap of infinite MPS with different unit cells")
            Ns = [self.get_B(i, 'B') for i in range(self.L)]
            Ms = [other.get_B(i, 'B') for i in range(self.L)]
            TM = TransferMatrix(Ns, Ms, transpose=False)
            leg_ket, leg_bra = Ms[-1].get_leg('vR'), Ns[-1].get_leg('vR')
            guess = npc.ones([leg_ket.conj(), leg_bra], dtype=np.result_type(self.dtype, other.dtype), labels=['vL', 'vL*'])
            if guess.stored_blocks == 0:
                return 0.
            val = None
            for _ in range(10):
                vals, vecs, N = Arnoldi(TM, guess, dict(N_min=2, N_max=30, P_tol=1.e-24, which='LM')).run()
                val, guess = vals[0], vecs[0]
                if N < 30:
                    break
            return val * self.norm * other.norm
        ov = MPSEnvironment(self, other).full_contraction(max(self.L // 2 - 1, 0))
        return ov * self.norm * other.norm

    def copy(self):
        """Copy sharing the (immutable) tensors: the engines replace tensors, they never modify them in place."""
        res = MPS(self.p_legs, list(self._B), list(self._S), form='B', bc=self.bc)
        res.form = list(self.form)
        res.norm = self.norm
        return res

    def compress_svd(self, trunc_par):
        """One right-sweep of QR decompositions without truncation, then a left-sweep of truncating SVDs (reference :5895).
        Returns the truncation error."""
        from ..linalg.truncation import svd_theta, TruncationError
        if not self.finite:
            raise NotImplementedError("tenpy_amd: compress_svd of infinite MPS")
        trunc_err = TruncationError()
        L = self.L
        B = self.get_B(0, 'Th')
        for i in range(L - 1):
            q, r = npc.qr(B.combine_legs(['vL', 'p']), inner_labels=['vR', 'vL'])
            self.set_B(i, q.split_legs(), form=None)
            B = npc.tensordot(r, self.get_B(i + 1, 'B'), axes=('vR', 'vL'))
        for i in range(L - 1, 0, -1):
            U, S, VH, err, norm_new = svd_theta(B.combine_legs(['p', 'vR']), trunc_par)
            trunc_err = trunc_err + err
            self.norm *= norm_new
            self.set_B(i, VH.split_legs(), form='B')
            B = npc.tensordot(self._B[i - 1], U, axes=('vR', 'vL')).iscale_axis(S, 'vR')
            self.set_SL(i, S)
        self.set_B(0, B, form='Th')
        return trunc_err

    def entanglement_entropy(self):
        res = []
        for s in (self._S[1:-1] if self.finite else self._S[:self.L]):      # infinite: the bond LEFT of every site
            if isinstance(s, npc.Array):      # bond matrix: its singular values are the Schmidt values
                _, s, _ = npc.svd(s, inner_labels=['vR', 'vL'])
                s = s / np.linalg.norm(s)
            p = s[s > 1e-30]**2
            res.append(float(-np.sum(p * np.log(p))))
        return np.array(res)

    def norm_test(self):
        """|<psi|psi>| computed by contracting the transfer matrices (device tensordots)."""
        B = self.get_B(0, 'B')
        E = npc.tensordot(B.conj(), B, axes=(['vL*', 'p*'], ['vL', 'p']))
        for i in range(1, self.L):
            B = self.get_B(i, 'B')
            E = npc.tensordot(E, B, axes=['vR', 'vL'])
            E = npc.tensordot(B.conj(), E, axes=(['vL*', 'p*'], ['vR*', 'p']))
        return E.to_ndarray().reshape(-1)[0]


class MPSEnvironment:
    """Partial contractions of ``<bra|ket>`` for two finite MPS (reference mps.py:6831): ``LP[i]`` (labels 'vR*', 'vR') is
    everything left of site i, ``RP[i]`` (labels 'vL', 'vL*') everything right of it.  Used by DMRG to orthogonalise
    against previously found states (``orthogonal_to``)."""

    def __init__(self, bra, ket):
        if bra.L != ket.L:
            raise ValueError("bra and ket must have the same length")
        self.bra, self.ket = bra, ket
        self.L = ket.L
        self.dtype = np.result_type(bra.dtype, ket.dtype)
        self._LP = [None] * self.L
        self._RP = [None] * self.L
        self._LP[0] = self._boundary(bra.get_B(0, None).get_leg('vL'), ket.get_B(0, None).get_leg('vL'), ['vR*', 'vR'])
        self._RP[self.L - 1] = self._boundary(ket.get_B(self.L - 1, None).get_leg('vR'),
                                              bra.get_B(self.L - 1, None).get_leg('vR'), ['vL', 'vL*'], ket_first=True)

    def _boundary(self, leg_a, leg_b, labels, ket_first=False):
        if leg_a.ind_len != 1 or leg_b.ind_len != 1:
            raise ValueError("finite MPS with trivial boundary legs expected")
        # LP: ('vR*' = bra's vL, 'vR' = conj of ket's vL);  RP: ('vL' = conj of ket's vR, 'vL*' = bra's vR)
        legs = [leg_a, leg_b.conj()] if not ket_first else [leg_a.conj(), leg_b]
        qt = legs[0].chinfo.make_valid(legs[0].get_charge(0) + legs[1].get_charge(0))
        return npc.Array.from_ndarray(np.ones((1, 1), dtype=self.dtype), legs, dtype=self.dtype, qtotal=qt, labels=labels)

    def get_LP(self, i, store=True):
        if self._LP[i] is not None:
            return self._LP[i]
        j = i
        while self._LP[j] is None:
            j -= 1
        LP = self._LP[j]
        for k in range(j, i):
            LP = self._contract_LP(k, LP)
            if store:
                self._LP[k + 1] = LP
        return LP

    def get_RP(self, i, store=True):
        if self._RP[i] is not None:
            return self._RP[i]
        j = i
        while self._RP[j] is None:
            j += 1
        RP = self._RP[j]
        for k in range(j, i, -1):
            RP = self._contract_RP(k, RP)
            if store:
                self._RP[k - 1] = RP
        return RP

    def del_LP(self, i):
        self._LP[i] = None

    def del_RP(self, i):
        self._RP[i] = None

    def _contract_LP(self, i, LP):
        LP = npc.tensordot(LP, self.ket.get_B(i, 'A'), axes=('vR', 'vL'))
        return npc.tensordot(self.bra.get_B(i, 'A').conj(), LP, axes=(['p*', 'vL*'], ['p', 'vR*']))     # 'vR*', 'vR'

    def _contract_RP(self, i, RP):
        RP = npc.tensordot(self.ket.get_B(i, 'B'), RP, axes=('vR', 'vL'))
        return npc.tensordot(RP, self.bra.get_B(i, 'B').conj(), axes=(['p', 'vL*'], ['p*', 'vR*']))     # 'vL', 'vL*'

    def full_contraction(self, i0):
        """``<bra|ket>`` evaluated with the environments around bond (i0, i0+1)."""
        LP = self.get_LP(i0 + 1, store=False)
        RP = self.get_RP(i0, store=False)
        S_bra, S_ket = self.bra.get_SR(i0), self.ket.get_SR(i0)
        LP = LP.scale_axis(S_ket, 'vR').scale_axis(np.conj(S_bra), 'vR*')
        return npc.inner(LP, RP, axes=(['vR*', 'vR'], ['vL*', 'vL']), do_conj=False)


class TransferMatrix:
    """Transfer matrix of the unit cell of two infinite MPS given by their tensor lists ``bra_N`` (conjugated inside) and
    ``ket_M`` (reference mps.py:6914).  ``transpose=False``: acts to the left on a right vector with labels 'vL', 'vL*';
    ``transpose=True``: acts to the right on a left vector with labels 'vR*', 'vR'.  A linear operator for ``Arnoldi``."""

    def __init__(self, bra_N, ket_M, transpose=False):
        self.transpose = transpose
        Ns = [N.conj() for N in bra_N]
        self._bra_N, self._ket_M = (list(reversed(Ns)), list(reversed(ket_M))) if not transpose else (Ns, list(ket_M))

    def matvec(self, vec):
        labels = vec.get_leg_labels()
        if not self.transpose:
            for N, M in zip(self._bra_N, self._ket_M):
                vec = npc.tensordot(M, vec, axes=['vR', 'vL'])
                vec = npc.tensordot(vec, N, axes=[['p', 'vL*'], ['p*', 'vR*']])
        else:
            for N, M in zip(self._bra_N, self._ket_M):
                vec = npc.tensordot(vec, M, axes=['vR', 'vL'])
                vec = npc.tensordot(N, vec, axes=[['vL*', 'p*'], ['vR*', 'p']])
        return vec if list(vec.get_leg_labels()) == labels else vec.transpose(labels)
