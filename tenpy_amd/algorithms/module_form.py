"""Device forms of the two callers that TeNPy's own engines construct per bond update, for the module form of the boundary
(``tenpy_amd.install.install(fused=True)``): the engines (``tenpy/algorithms/dmrg.py:529 update_local``,
``mps_common.py:516 make_eff_H``) stay the reference's code; only what they instantiate is rebound.

* ``device_two_site_h(RefTwoSiteH)`` -> a class with the constructor and interface of the reference's ``TwoSiteH``
  (``mps_common.py:1245``: ``(env, i0, combine=False, move_right=True)``, ``matvec`` :1321, ``combine_theta`` :1374,
  ``update_LP`` / ``update_RP`` :1421-1437, ``LHeff`` / ``RHeff`` for the mixers :1887-1898) on top of
  ``tenpy_amd.algorithms.mps_common.TwoSiteH``: cached contraction plans, fused ``LHeff`` build, and -- for
  ``combine=False``, the reference's default -- the factored matvec ``LP . theta . (W0 W1) . RP`` with the MPO tensors
  applied as block-level linear combinations instead of K = 1 GEMMs.  Bonds the device form does not cover (small sectors,
  MPOs with non-scalar blocks, ``H + h.c.`` environments, exact diagonalisation of small bonds) get the reference's own
  class: ``__new__`` dispatches, so the engines see one ``EffectiveH``.
* ``hinted_mixed_svd(ref_mixed_svd)`` wraps ``TwoSiteDMRGEngine.mixed_svd`` (``dmrg.py:876``) to tell the block SVD which
  bond it decomposes (``np_conserved.svd_hint``), which enables the warm start of ``linalg/_svd_warm.py``.
"""
import numpy as np

from ..linalg import np_conserved as npc
from . import mps_common as dev_mc

__all__ = ['device_two_site_h', 'hinted_mixed_svd']

# Smallest "largest bond sector" for which the device form is used.  Round 3 measurement (module form on the MI355X, Heisenberg
# L = 100, mixer ramp): with the reference's own class below 64 -- four generic tensordots with transposed copies per matvec,
# ~20 device calls and ~1 ms of interpreter time each -- a chi <= 256 sweep took 4.4 - 4.9 s against 0.9 - 1.5 s with the
# device form everywhere, so the threshold is 1 (kept as a knob for A/B runs).
MIN_SECTOR = 1
stats = {'device': 0, 'reference': 0}      # bonds handled by the device form / handed back to the reference's class


def device_two_site_h(Ref):
    class DeviceTwoSiteH(dev_mc.TwoSiteH):
        __doc__ = "Device form of tenpy.algorithms.mps_common.TwoSiteH (see tenpy_amd/algorithms/module_form.py)."
        length = 2
        acts_on = ['vL', 'p0', 'p1', 'vR']
        _reference_class = Ref

        def __new__(cls, env, i0, combine=False, move_right=True):
            if cls._device_ok(env, i0, combine):
                stats['device'] += 1
                return object.__new__(cls)
            stats['reference'] += 1
            return Ref(env, i0, combine, move_right)          # not an instance of cls: __init__ below is skipped

        @staticmethod
        def _device_ok(env, i0, combine):
            try:
                if getattr(env, 'has_hc', False) or getattr(env.H, 'explicit_plus_hc', False):
                    return False
                LP, RP = env.get_LP(i0), env.get_RP(i0 + 1)
                if int(np.max(LP.get_leg('vR').get_block_sizes())) < MIN_SECTOR:
                    return False
                if sorted(LP.get_leg_labels()) != sorted(['vR*', 'wR', 'vR']) or sorted(RP.get_leg_labels()) != sorted(['wL', 'vL', 'vL*']):
                    return False
                if not combine:         # factored form: every block of W0, W1 a single number, no transposed copies needed
                    return dev_mc.factored_matvec_possible(LP, RP, env.H.get_W(i0), env.H.get_W(i0 + 1))
                return True
            except Exception:
                return False

        def __init__(self, env, i0, combine=False, move_right=True):
            dev_mc.TwoSiteH.__init__(self, env, i0, combine=True, move_right=move_right, factored=not combine)
            if (not combine) and not self.factored:
                raise RuntimeError("tenpy_amd: factored matvec not applicable although _device_ok said so")
            self.combine = combine
            self.acts_on = ['(vL.p0)', '(p1.vR)'] if combine else ['vL', 'p0', 'p1', 'vR']

        def combine_theta(self, theta):
            theta = dev_mc.TwoSiteH.combine_theta(self, theta)
            return theta.itranspose(self.acts_on) if list(theta.get_leg_labels()) != self.acts_on else theta

        def matvec(self, theta):
            labels = theta.get_leg_labels()
            res = dev_mc.TwoSiteH.matvec(self, theta)
            if list(res.get_leg_labels()) != list(labels):
                res.itranspose(labels)
            return res

        def update_LP(self, env, i, U=None):
            if U is None or not (self.combine or self.factored):
                return env.get_LP(i, store=True)
            lab = '(vL.p)' if U.has_label('(vL.p)') else '(vL.p0)'
            if not U.has_label(lab):
                return env.get_LP(i, store=True)
            return dev_mc.TwoSiteH.update_LP(self, env, i, U.replace_label(lab, '(vL.p0)'))

        def update_RP(self, env, i, VH=None):
            if VH is None or not (self.combine or self.factored):
                return env.get_RP(i, store=True)
            lab = '(p.vR)' if VH.has_label('(p.vR)') else '(p1.vR)'
            if not VH.has_label(lab):
                return env.get_RP(i, store=True)
            return dev_mc.TwoSiteH.update_RP(self, env, i, VH.replace_label(lab, '(p1.vR)'))

        def to_matrix(self):
            """Contract to a matrix Array like the reference (:1392)."""
            if self.combine:
                return self.to_matrix_array()
            contr = npc.tensordot(self.LP, self.W0, axes=['wR', 'wL'])
            contr = npc.tensordot(contr, self.W1, axes=['wR', 'wL'])
            contr = npc.tensordot(contr, self.RP, axes=['wR', 'wL'])
            return contr.combine_legs([['vR*', 'p0', 'p1', 'vL*'], ['vR', 'p0*', 'p1*', 'vL']], qconj=[+1, -1])

        def adjoint(self):
            raise NotImplementedError("DeviceTwoSiteH is not used for H + h.c. environments (see _device_ok)")

    DeviceTwoSiteH.__name__ = 'TwoSiteH'
    DeviceTwoSiteH.__qualname__ = 'TwoSiteH'
    return DeviceTwoSiteH


def hinted_mixed_svd(ref_mixed_svd):
    def mixed_svd(self, theta):
        if self.mixer is None:       # plain svd_theta: the next npc.svd decomposes the theta of bond (i0, i0 + 1)
            from ..linalg import _svd_warm
            npc.svd_hint = ((_svd_warm.owner_token(self.psi), int(self.i0)), 'R' if self.move_right else 'L')
        try:
            return ref_mixed_svd(self, theta)
        finally:
            npc.svd_hint = None
    mixed_svd.__doc__ = ref_mixed_svd.__doc__
    mixed_svd._tpa_wrapped = True
    return mixed_svd
