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
import os
import warnings

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


def batched_tebd_evolve_step(ref_tebd):
    """Fused caller for the reference's ``TEBDEngine.evolve_step`` (algorithms/tebd.py:374-414): the bonds of one half-step do not
    share a site, so their decompositions go to the device as ONE batched call each (``np_conserved.svd_batched`` for
    ``TEBDEngine.update_bond``, tebd.py:416-483; ``qr_batched`` + ``svd_batched`` / ``eigh_batched`` for ``QRBasedTEBDEngine.update_bond``, :685-738).
    The per-bond statements before and after the decomposition are the reference's, in the reference's order; engines whose
    ``update_bond`` is not one of those two (subclasses), and everything else, run the reference's loop; with ``use_eig_based_svd``
    the Hermitian eigenproblems of the bond matrices are ONE ``eigh_batched`` call."""
    import numpy as np
    from ..linalg import truncation as dev_trunc
    ref_evolve_step = ref_tebd.TEBDEngine.evolve_step
    ref_update = ref_tebd.TEBDEngine.update_bond
    ref_update_qr = ref_tebd.QRBasedTEBDEngine.update_bond
    TruncationError = ref_tebd.TruncationError

    def evolve_step(self, U_idx_dt, odd):
        upd = type(self).update_bond
        qr_based = upd is ref_update_qr
        if not (upd is ref_update or qr_based) or not getattr(self.psi, 'finite', False):
            return ref_evolve_step(self, U_idx_dt, odd)
        Us = self._U[U_idx_dt]
        bonds = [int(i) for i in np.arange(int(odd) % 2, self.psi.L, 2) if Us[i] is not None]
        if len(bonds) < 2:
            return ref_evolve_step(self, U_idx_dt, odd)
        psi = self.psi
        # groups of bonds whose work areas fit the device together (algorithms/tebd.batch_group_size: ~24x the dense theta each)
        from .tebd import batch_bytes_cap
        cap = batch_bytes_cap()
        worst = max(24 * 16 * (psi.sites[i - 1].dim * len(psi.get_SL(i - 1))) * (psi.sites[i].dim * len(psi.get_SR(i))) for i in bonds)
        group = int(max(1, min(len(bonds), cap // max(worst, 1))))
        total = TruncationError()
        for g0 in range(0, len(bonds), group):
            total += _half_step_group(self, bonds[g0:g0 + group], Us, qr_based)
        self._update_index = None
        return total

    def _half_step_group(self, bonds, Us, qr_based):
        psi = self.psi
        Cs, thetas, extra = [], [], []
        for i in bonds:
            i0, i1 = i - 1, i
            C = psi.get_theta(i0, n=2, formL=0.0)
            C = npc.tensordot(Us[i], C, axes=(['p0*', 'p1*'], ['p0', 'p1']))
            C.itranspose(['vL', 'p0', 'p1', 'vR'])
            theta = C.scale_axis(psi.get_SL(i0), 'vL')
            theta = theta.combine_legs([('vL', 'p0'), ('p1', 'vR')], qconj=[+1, -1])
            Cs.append(C)
            thetas.append(theta)
            if qr_based:
                old_B_L, old_B_R = psi.get_B(i0, 'B'), psi.get_B(i1, 'B')
                extra.append((old_B_L.qtotal, old_B_R.qtotal, old_B_R.get_leg('vL'), theta, False, self._expansion_rate(i),
                              self.options.get('cbe_min_block_increase', 1, int)))
            else:
                extra.append([psi.get_B(i0, None).qtotal, None])
        total = TruncationError()
        if qr_based:
            compute_err = self.options.get('compute_err', True, bool)
            res = dev_trunc.decompose_theta_qr_based_batched(extra, self.trunc_params, compute_err, False,
                                                             use_eig_based_svd=self.options.get('use_eig_based_svd', False, bool))
            for i, C, theta, (_, S, B_R, form, err, renormalize) in zip(bonds, Cs, thetas, res):
                i0, i1 = i - 1, i
                if compute_err:         # the reference's warning (tebd.py:712-725), same condition, same text
                    chi_max = self.trunc_params.get('chi_max', None)
                    if err.eps > 1e-16 and chi_max is not None and len(S) < chi_max:
                        warnings.warn('QRBased decomposition resulted in large truncation error even though the bond '
                                      'dimension is not maxed out yet. Try increasing the expansion rate, e.g. '
                                      'via the `cbe_expand_0` and `cbe_min_block_increase` options for the engine. '
                                      'You probably should compare to a "regular" (non QR-based) engine and see how '
                                      'fast the bond dimension needs to grow for your scenario. '
                                      'See https://github.com/tenpy/tenpy/pull/513 .', stacklevel=2)
                assert form[1] == 'B'
                err = TruncationError(err.eps, err.ov)
                B_L = npc.tensordot(C.combine_legs(('p1', 'vR'), pipes=theta.legs[1]), B_R.conj(), axes=[['(p1.vR)'], ['(p*.vR*)']]) / renormalize
                B_L.ireplace_labels(['p0', 'vL*'], ['p', 'vR'])
                B_R = B_R.split_legs(1)
                psi.norm *= renormalize
                psi.set_B(i0, B_L, form='B')
                psi.set_SL(i1, S)
                psi.set_B(i1, B_R, form='B')
                self._trunc_err_bonds[i] = self._trunc_err_bonds[i] + err
                total += err
        else:
            npc.svd_engine_floor = True
            res = dev_trunc.svd_theta_batched(thetas, self.trunc_params, extra, inner_labels=['vR', 'vL'])
            for i, C, theta, (U, S, V, err, renormalize) in zip(bonds, Cs, thetas, res):
                i0, i1 = i - 1, i
                err = TruncationError(err.eps, err.ov)
                B_R = V.split_legs(1).ireplace_label('p1', 'p')
                B_L = npc.tensordot(C.combine_legs(('p1', 'vR'), pipes=theta.legs[1]), V.conj(), axes=['(p1.vR)', '(p1*.vR*)'])
                B_L.ireplace_labels(['vL*', 'p0'], ['vR', 'p'])
                B_L /= renormalize
                psi.norm *= renormalize
                psi.set_SR(i0, S)
                psi.set_B(i0, B_L, form='B')
                psi.set_B(i1, B_R, form='B')
                self._trunc_err_bonds[i] = self._trunc_err_bonds[i] + err
                total += err
        return total

    evolve_step.__doc__ = ref_evolve_step.__doc__
    evolve_step._tpa_wrapped = True
    return evolve_step
