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

__all__ = ['TwoSiteH']


class TwoSiteH:
    length = 2
    acts_on = ['(vL.p0)', '(p1.vR)']

    def __init__(self, env, i0, combine=True, move_right=True):
        if not combine:
            raise NotImplementedError("tenpy_amd.TwoSiteH: only combine=True")
        self.i0 = i0
        self.combine = combine
        self.move_right = move_right
        self.LP = env.get_LP(i0)
        self.RP = env.get_RP(i0 + 1)
        self.W0 = env.H.get_W(i0).replace_labels(['p', 'p*'], ['p0', 'p0*'])
        self.W1 = env.H.get_W(i0 + 1).replace_labels(['p', 'p*'], ['p1', 'p1*'])
        self.dtype = env.H.dtype
        self.combine_Heff()
        self._plans = None
        self.N = self.pipeL.ind_len * self.pipeR.ind_len
        self.flops_per_matvec = None
        self.bytes_per_matvec = None

    def combine_Heff(self):
        """LHeff = LP.W0 and RHeff = W1.RP with the (virtual, physical) legs fused into pipes."""
        LHeff = npc.tensordot(self.LP, self.W0, axes=['wR', 'wL'])        # vR*, vR, wR, p0, p0*
        self.pipeL = pipeL = LHeff.make_pipe(['vR*', 'p0'], qconj=+1)
        self.LHeff = LHeff.combine_legs([['vR*', 'p0'], ['vR', 'p0*']], pipes=[pipeL, pipeL.conj()],
                                        new_axes=[0, 2])                   # (vR*.p0), wR, (vR.p0*)
        RHeff = npc.tensordot(self.W1, self.RP, axes=['wR', 'wL'])        # wL, p1, p1*, vL, vL*
        self.pipeR = pipeR = RHeff.make_pipe(['p1', 'vL*'], qconj=-1)
        self.RHeff = RHeff.combine_legs([['p1*', 'vL'], ['p1', 'vL*']], pipes=[pipeR.conj(), pipeR],
                                        new_axes=[1, 2])                   # wL, (p1*.vL), (p1.vL*)

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
