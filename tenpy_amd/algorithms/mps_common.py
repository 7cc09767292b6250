"""Effective two-site Hamiltonian: the Lanczos matvec of the hot path.

Mirrors ``TwoSiteH`` of ``tenpy/algorithms/mps_common.py`` (:1245; ``matvec`` :1321-1348, ``combine_Heff``
:1350, ``combine_theta`` :1374, ``update_LP`` :1421, ``update_RP`` :1430) for ``combine=True``:

    |theta'> = LHeff . theta . RHeff          (two block-sparse tensordots)

MI355X-first differences:
* the leg order of ``RHeff`` is chosen so that NEITHER tensordot of the matvec needs a transpose
  (the reference calls ``itranspose`` inside ``matvec`` :1339): LHeff = [(vR*.p0), wR, (vR.p0*)],
  RHeff = [wL, (p1*.vL), (p1.vL*)];
* both contraction plans are built once per bond and replayed for every Lanczos step: one matvec is
  exactly two grouped-GEMM launches, no host planning, no allocation besides the result arena.
"""
import numpy as np

from ..linalg import np_conserved as npc

__all__ = ['TwoSiteH', 'DensityMatrixMixer']


FUSED_HEFF = True     # tuning / test hook: False forces the generic tensordot + combine_legs construction

# ---------------------------------------------------------------------------------------------------------------------
# Fused construction of LHeff / RHeff:  LP.W0 (resp. W1.RP) + combine_legs in ONE kernel launch.
# When every charge block of the MPO tensor is a single number (all charges resolved: spin-1/2 with Sz, Hubbard with
# (N, Sz), parity-conserving TFI ...), ``tensordot(LP, W0, 'wR'-'wL')`` is a sum of scaled copies of the LP[:, w, :]
# blocks -- as a GEMM it has inner dimension 1 per link (64072 almost empty 64 x 64 tiles for a chi = 2048 Heisenberg
# bond, 1.4 ms) and is followed by a 131 MB repacking copy (combine_legs, 0.2 ms).  ``tpa_lincomb_batch`` writes every
# (w', p, p*) slab of the fused tensor directly at its place inside the pipe blocks.
def _mpo_entries(W):
    """(qdata, values) of an MPO tensor whose stored blocks are all 1 x 1 x 1 x 1, else ``None`` (cached on W)."""
    ent = getattr(W, '_tpa_entries', False)
    if ent is False:
        if all(np.all(leg.get_block_sizes() == 1) for leg in W.legs) and W.stored_blocks > 0:
            vals = np.array([np.asarray(b).reshape(-1)[0] for b in W._data])
            ent = (np.array(W._qdata), vals)
        else:
            ent = None
        W._tpa_entries = ent
    return ent


def _lincomb_into(dst, src, jobs, terms, max_elems):
    from ..linalg import _device as dev
    if len(jobs) == 0:
        return
    L = dev.lib()
    terms_d = dev.to_device(np.ascontiguousarray(terms))
    for s0 in range(0, len(jobs), 60000):
        jd = dev.to_device(np.ascontiguousarray(jobs[s0:s0 + 60000]))
        dev.check(L.tpa_lincomb_batch(dev.code(dst.dtype), jd.data_ptr(), len(jobs[s0:s0 + 60000]), terms_d.data_ptr(),
                                      int(max_elems), src._arena.data_ptr(), dst._arena.data_ptr(), dev.stream()), "lincomb")


def _fused_heff(env_t, W, left):
    """``left``: LHeff [(vR*.p0), wR, (vR.p0*)] from LP [vR*, wR, vR] and W0 [wL, wR, p0, p0*];
    else RHeff [wL, (p1*.vL), (p1.vL*)] from RP [wL, vL, vL*] and W1 [wL, wR, p1, p1*].
    Returns (Heff, pipe) or ``None`` when the fast path does not apply (generic tensordot + combine_legs then)."""
    from ..linalg import _device as dev
    from ..linalg.charges import LegPipe
    ent = _mpo_entries(W)
    pl = 'p0' if left else 'p1'
    want_env = ['vR*', 'wR', 'vR'] if left else ['wL', 'vL', 'vL*']
    if ent is None or list(env_t.get_leg_labels()) != want_env or list(W.get_leg_labels()) != ['wL', 'wR', pl, pl + '*'] \
            or env_t.dtype != np.result_type(env_t.dtype, W.dtype) or env_t.stored_blocks == 0:
        return None
    wq, wv = ent
    w_axis_env = 1 if left else 0
    if not np.all(env_t.legs[w_axis_env].get_block_sizes() == 1):
        return None
    eq = env_t._qdata
    shapes = env_t._block_shapes()
    if left:
        pipe = LegPipe([env_t.get_leg('vR*'), W.get_leg('p0')], qconj=+1)
        legs = [pipe, W.get_leg('wR'), pipe.conj()]
        labels = ['(vR*.p0)', 'wR', '(vR.p0*)']
    else:
        pipe = LegPipe([W.get_leg('p1'), env_t.get_leg('vL*')], qconj=-1)
        legs = [W.get_leg('wL'), pipe.conj(), pipe]
        labels = ['wL', '(p1*.vL)', '(p1.vL*)']
    qm = pipe.q_map
    n0, n1 = pipe.legs[0].block_number, pipe.legs[1].block_number
    row_of = np.full((n0, n1), -1, dtype=np.int64)
    row_of[qm[:, 3], qm[:, 4]] = np.arange(len(qm))
    psz = pipe.get_block_sizes()
    # all (env block, W entry) pairs that share the contracted MPO index
    wc = wq[:, 0] if left else wq[:, 1]            # W index contracted with the environment
    ie, iw = np.nonzero(eq[:, w_axis_env][:, None] == wc[None, :])
    if len(ie) == 0:
        return None
    if left:      # rows (vR*, p0), cols (vR, p0*)
        r_row = row_of[eq[ie, 0], wq[iw, 2]]
        c_row = row_of[eq[ie, 2], wq[iw, 3]]
        w_out = wq[iw, 1]
        rows, cols = shapes[ie, 0], shapes[ie, 2]
    else:         # rows (p1*, vL), cols (p1, vL*)
        r_row = row_of[wq[iw, 3], eq[ie, 1]]
        c_row = row_of[wq[iw, 2], eq[ie, 2]]
        w_out = wq[iw, 0]
        rows, cols = shapes[ie, 1], shapes[ie, 2]
    Qr, Qc = qm[r_row, 2], qm[c_row, 2]
    r0, c0 = qm[r_row, 0], qm[c_row, 0]
    bq = np.stack([Qr, w_out, Qc], axis=1) if left else np.stack([w_out, Qr, Qc], axis=1)
    ubq, inv = np.unique(bq, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    order = np.lexsort(ubq.T)                       # last leg most significant, like every sorted _qdata
    rank_of = np.empty(len(order), dtype=np.int64)
    rank_of[order] = np.arange(len(order))
    ubq = ubq[order]
    blk = rank_of[inv]
    res = npc.Array(legs, env_t.dtype, env_t.chinfo.make_valid(env_t.qtotal + W.qtotal), labels)
    sizes = res._set_blocks(ubq, zero=True, qdata_sorted=True)
    ld = psz[Qc]
    dst_off = res._offsets[blk] + r0 * ld + c0
    # one job per destination slab, its terms = all contributions landing there (deterministic order)
    key = np.stack([dst_off, np.arange(len(dst_off))], axis=1)
    perm = np.lexsort((key[:, 1], key[:, 0]))
    d_sorted = dst_off[perm]
    first = np.concatenate([[True], d_sorted[1:] != d_sorted[:-1]])
    starts = np.nonzero(first)[0]
    counts = np.diff(np.concatenate([starts, [len(perm)]]))
    jobs = np.zeros((len(starts), 8), dtype=np.int64)
    pf = perm[starts]
    jobs[:, 0], jobs[:, 1], jobs[:, 2], jobs[:, 3] = d_sorted[starts], rows[pf], cols[pf], ld[pf]
    jobs[:, 4], jobs[:, 5] = starts, counts
    terms = np.zeros((len(perm), 4), dtype=np.int64)
    terms[:, 0] = env_t._offsets[ie[perm]]
    terms[:, 1] = shapes[ie[perm], 2]
    alpha = np.asarray(wv[iw[perm]], dtype=np.complex128)
    terms[:, 2] = np.ascontiguousarray(alpha.real).view(np.int64)
    terms[:, 3] = np.ascontiguousarray(alpha.imag).view(np.int64)
    _lincomb_into(res, env_t, jobs, terms, int(np.max(rows * cols)))
    return res, pipe


class TwoSiteH:
    length = 2
    acts_on = ['(vL.p0)', '(p1.vR)']

    def __init__(self, env, i0, combine=True, move_right=True, tensors=None):
        if not combine:
            raise NotImplementedError("tenpy_amd.TwoSiteH: only combine=True")
        self.i0 = i0
        self.combine = combine
        self.move_right = move_right
        if tensors is not None:      # explicit (LP, RP, W0, W1), e.g. replaying a dumped bond
            self.LP, self.RP, W0, W1 = tensors
            self.dtype = W0.dtype
        else:
            self.LP = env.get_LP(i0)
            self.RP = env.get_RP(i0 + 1)
            W0, W1 = env.H.get_W(i0), env.H.get_W(i0 + 1)
            self.dtype = env.H.dtype
        self.W0 = W0.replace_labels(['p', 'p*'], ['p0', 'p0*'])
        self.W1 = W1.replace_labels(['p', 'p*'], ['p1', 'p1*'])
        self.combine_Heff(env if tensors is None else None)
        self._plans = None
        self.N = self.pipeL.ind_len * self.pipeR.ind_len
        self.flops_per_matvec = None
        self.bytes_per_matvec = None

    def combine_Heff(self, env=None):
        """LHeff = LP.W0 and RHeff = W1.RP with the (virtual, physical) legs fused into pipes.

        Reference: ``TwoSiteH.combine_Heff`` (mps_common.py:1350).  With an environment, the fused tensors are kept in
        ``env._heff_cache`` keyed by (side, site) and reused as long as the environment tensor they were built from is
        still the stored one: in a right-moving half sweep RHeff of a bond is exactly the one built when the previous
        left-moving half sweep optimised that bond (and vice versa), so only one of the two is rebuilt per update.
        All cached tensors stay in HBM (2 x 131 MB per bond at chi=2048, < 26 GB for L=100)."""
        cache = getattr(env, '_heff_cache', None) if env is not None else None
        hit = cache.get(('L', self.i0)) if cache is not None else None
        if hit is not None and hit[0] is self.LP:
            _, self.LHeff, self.pipeL = hit
        else:
            fused = _fused_heff(self.LP, self.W0, True) if FUSED_HEFF else None
            if fused is not None:
                self.LHeff, self.pipeL = fused
            else:
                LHeff = npc.tensordot(self.LP, self.W0, axes=['wR', 'wL'])        # vR*, vR, wR, p0, p0*
                self.pipeL = pipeL = LHeff.make_pipe(['vR*', 'p0'], qconj=+1)
                self.LHeff = LHeff.combine_legs([['vR*', 'p0'], ['vR', 'p0*']], pipes=[pipeL, pipeL.conj()],
                                                new_axes=[0, 2])                   # (vR*.p0), wR, (vR.p0*)
            if cache is not None:
                cache[('L', self.i0)] = (self.LP, self.LHeff, self.pipeL)
        hit = cache.get(('R', self.i0 + 1)) if cache is not None else None
        if hit is not None and hit[0] is self.RP:
            _, self.RHeff, self.pipeR = hit
        else:
            fused = _fused_heff(self.RP, self.W1, False) if FUSED_HEFF else None
            if fused is not None:
                self.RHeff, self.pipeR = fused
            else:
                RHeff = npc.tensordot(self.W1, self.RP, axes=['wR', 'wL'])        # wL, p1, p1*, vL, vL*
                self.pipeR = pipeR = RHeff.make_pipe(['p1', 'vL*'], qconj=-1)
                self.RHeff = RHeff.combine_legs([['p1*', 'vL'], ['p1', 'vL*']], pipes=[pipeR.conj(), pipeR],
                                                new_axes=[1, 2])                   # wL, (p1*.vL), (p1.vL*)
            if cache is not None:
                cache[('R', self.i0 + 1)] = (self.RP, self.RHeff, self.pipeR)

    def combine_theta(self, theta):
        """theta (vL, p0, p1, vR) -> matrix [(vL.p0), (p1.vR)] using the pipes of Heff."""
        return theta.combine_legs([['vL', 'p0'], ['p1', 'vR']], pipes=[self.pipeL, self.pipeR])

    def matvec(self, theta):
        """theta [(vL.p0), (p1.vR)] -> H_eff theta, same legs and labels."""
        if self._plans is None or not self._plan_matches(theta):
            p1, l_use, t_use = npc.plan_tensordot(self.LHeff, theta, axes=['(vR.p0*)', '(vL.p0)'])
            assert l_use is self.LHeff and t_use is theta, "matvec step 1 must not need a transpose"
            tmp = p1.apply(self.LHeff, theta)
            p2, t2_use, r_use = npc.plan_tensordot(tmp, self.RHeff, axes=(['wR', '(p1.vR)'], ['wL', '(p1*.vL)']))
            assert t2_use is tmp and r_use is self.RHeff, "matvec step 2 must not need a transpose"
            self._plans = (p1, p2, theta._struct_key(), theta.dtype)
            if not p1.empty and not p2.empty:
                self.flops_per_matvec = p1.flops + p2.flops
                self.bytes_per_matvec = p1.bytes_min + p2.bytes_min
                self.gemm_shapes = (p1.gemm_shapes, p2.gemm_shapes)
            res = p2.apply(tmp, self.RHeff)
        else:
            p1, p2 = self._plans[0], self._plans[1]
            tmp = p1.apply(self.LHeff, theta)
            res = p2.apply(tmp, self.RHeff)
        res.iset_leg_labels(['(vL.p0)', '(p1.vR)'])
        return res

    def _plan_matches(self, theta):
        return self._plans[2] == theta._struct_key() and self._plans[3] == theta.dtype

    def update_LP(self, env, i, U=None):
        """LP(i0+1) = U^dagger LHeff U   (reference :1421)."""
        assert i == self.i0 + 1
        LP = npc.tensordot(self.LHeff, U, axes=['(vR.p0*)', '(vL.p0)'])
        LP = npc.tensordot(U.conj(), LP, axes=['(vL*.p0*)', '(vR*.p0)'])      # vR*, wR, vR
        env.set_LP(i, LP)
        return LP

    def update_RP(self, env, i, VH=None):
        """RP(i0) = RHeff VH VH^dagger   (reference :1430)."""
        assert i == self.i0
        RP = npc.tensordot(VH, self.RHeff, axes=['(p1.vR)', '(p1*.vL)'])      # vL, wL, (p1.vL*)
        RP = npc.tensordot(RP, VH.conj(), axes=['(p1.vL*)', '(p1*.vR*)'])     # vL, wL, vL*
        env.set_RP(i, RP)
        return RP

    def to_matrix(self):
        """Dense effective Hamiltonian on the host (tests only)."""
        L, R = self.LHeff.to_ndarray(), self.RHeff.to_ndarray()
        full = np.tensordot(L, R, axes=([1], [0]))           # (vR*.p0), (vR.p0*), (p1*.vL), (p1.vL*)
        full = full.transpose(0, 3, 1, 2)                    # out_L, out_R, in_L, in_R
        n = full.shape[0] * full.shape[1]
        return full.reshape(n, n)


class DensityMatrixMixer:
    """Density-matrix perturbation ("mixer") of two-site DMRG -- the npc call sequence of the reference's
    ``DensityMatrixMixer.mix_rho`` / ``svd_from_rho`` (mps_common.py:1972-2079): four tensordots with the
    effective Hamiltonian halves, ``iscale_axis`` on the MPO leg and two block ``eigh``.

    ``IdL`` / ``IdR`` are the MPO indices on the bond (i0, i0+1) meaning "only identities to the left / right"
    (``_mix_LR``, :1846).  Returns a general (non-diagonal) bond matrix ``S`` like the reference.
    """

    def __init__(self, amplitude=1.e-5, IdL=0, IdR=-1, explicit_plus_hc=False, decay=2., disable_after=15,
                 sweep_activated=0):
        assert amplitude <= 1.
        self.amplitude = amplitude
        self.IdL, self.IdR = IdL, IdR
        self.explicit_plus_hc = explicit_plus_hc
        self.decay, self.disable_after, self.sweep_activated = decay, disable_after, sweep_activated

    def update_amplitude(self, sweeps):
        """Divide the amplitude by ``decay`` after a sweep; returns ``None`` when the mixer should be switched off
        (``disable_after`` sweeps after activation or amplitude below machine precision; reference :1626-1653)."""
        off = self.disable_after is not None and sweeps >= self.sweep_activated + self.disable_after
        if self.amplitude is not None and self.decay is not None:
            self.amplitude /= self.decay
            if self.amplitude <= np.finfo('float').eps:
                off = True
        return None if off else self

    def _mix_LR(self, chi_MPO):
        mix_L = np.full((chi_MPO,), self.amplitude)
        mix_R = np.full((chi_MPO,), self.amplitude)
        one = 1. if not self.explicit_plus_hc else 0.5
        if self.IdL is not None:
            mix_L[self.IdL] = one
            mix_R[self.IdL] = 0.
        if self.IdR is not None:
            mix_L[self.IdR] = 0.
            mix_R[self.IdR] = one
        return mix_L, mix_R

    def mix_rho(self, eff_H, theta, mix_left, mix_right):
        """``rho_L`` [(vL.p0), (vL*.p0*)] and ``rho_R`` [(p1.vR), (p1*.vR*)], perturbed with H where requested."""
        chi_MPO = eff_H.LHeff.get_leg('wR').ind_len
        mix_L, mix_R = self._mix_LR(chi_MPO)
        if mix_left:
            rho_L = npc.tensordot(eff_H.LHeff, theta, axes=['(vR.p0*)', '(vL.p0)'])
            rho_L.ireplace_label('(vR*.p0)', '(vL.p0)')
            rho_c = rho_L.conj()
            rho_L.iscale_axis(mix_L, 'wR')
            rho_L = npc.tensordot(rho_L, rho_c, axes=[['wR', '(p1.vR)'], ['wR*', '(p1*.vR*)']])
            if self.explicit_plus_hc:
                rho_L = rho_L + rho_L.conj().itranspose()
            if self.IdL is None:
                rho_L = rho_L + npc.tensordot(theta, theta.conj(), axes=['(p1.vR)', '(p1*.vR*)'])
        else:
            rho_L = npc.tensordot(theta, theta.conj(), axes=['(p1.vR)', '(p1*.vR*)'])
        if mix_right:
            # RHeff is stored as [wL, (p1*.vL), (p1.vL*)] (matvec-friendly order)
            rho_R = npc.tensordot(theta, eff_H.RHeff, axes=['(p1.vR)', '(p1*.vL)'])
            rho_R.ireplace_label('(p1.vL*)', '(p1.vR)')
            rho_c = rho_R.conj()
            rho_R.iscale_axis(mix_R, 'wL')
            rho_R = npc.tensordot(rho_c, rho_R, axes=[['wL*', '(vL*.p0*)'], ['wL', '(vL.p0)']])
            if self.explicit_plus_hc:
                rho_R = rho_R + rho_R.conj().itranspose()
            if self.IdR is None:
                rho_R = rho_R + npc.tensordot(theta.conj(), theta, axes=['(vL*.p0*)', '(vL.p0)'])
        else:
            rho_R = npc.tensordot(theta.conj(), theta, axes=['(vL*.p0*)', '(vL.p0)'])
        return rho_L, rho_R

    def svd_from_rho(self, rho_L, rho_R, theta, trunc_params, qtotal_LR=None):
        """Diagonalise rho_L / rho_R and rewrite theta = U S VH with isometric U, VH and a bond MATRIX S."""
        from ..linalg.truncation import truncate
        chinfo = theta.chinfo
        qL, qR = (None, None) if qtotal_LR is None else qtotal_LR
        if qL is None and qR is None:
            qL, qR = chinfo.make_valid(), theta.qtotal
        elif qL is None:
            qL = chinfo.make_valid(theta.qtotal - qR)
        elif qR is None:
            qR = chinfo.make_valid(theta.qtotal - qL)
        rho_L = rho_L.transpose(['(vL.p0)', '(vL*.p0*)'])
        rho_R = rho_R.transpose(['(p1.vR)', '(p1*.vR*)'])
        val_L, U = npc.eigh(rho_L)
        U.iset_leg_labels(['(vL.p0)', 'vR'])
        val_L[val_L < 0.] = 0.
        val_L /= np.sum(val_L)
        S_a = np.sqrt(val_L)
        keep_L, _, err_L = truncate(S_a, trunc_params)
        U.iproject(keep_L, axes='vR')
        U = U.gauge_total_charge(1, qL)
        val_R, Vc = npc.eigh(rho_R)
        Vc.iset_leg_labels(['(p1.vR)', 'vL'])
        VH = Vc.itranspose(['vL', '(p1.vR)'])
        val_R[val_R < 0.] = 0.
        val_R /= np.sum(val_R)
        keep_R, _, err_R = truncate(np.sqrt(val_R), trunc_params)
        VH.iproject(keep_R, axes='vL')
        VH = VH.gauge_total_charge(0, qR)
        S = npc.tensordot(U.conj(), theta, axes=['(vL*.p0*)', '(vL.p0)'])
        S = npc.tensordot(S, VH.conj(), axes=['(p1.vR)', '(p1*.vR*)'])
        S.ireplace_labels(['vR*', 'vL*'], ['vL', 'vR'])
        S.iscale_prefactor(1. / S.norm())
        return U, S, VH, err_L + err_R, S_a[keep_L]
