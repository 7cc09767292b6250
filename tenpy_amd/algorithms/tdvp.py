"""Time-dependent variational principle for finite MPS -- the other time-evolution caller of the hot path.

Mirrors ``tenpy/algorithms/tdvp.py`` (``TDVPEngine`` :58, ``TwoSiteTDVPEngine`` :233, ``SingleSiteTDVPEngine`` :318): per local
update one Krylov exponential ``exp(-i dt/2 H_eff) theta`` (``LanczosEvolution``: the same matvec + fused vector update as the
DMRG ground-state search), a block SVD (truncating for two sites, exact for one site) and a backward step with the
effective Hamiltonian of one site less.  Sweep schedules, time-step conventions (last update of the right sweep gets the
doubled step) and the returned truncation error follow the reference.  No basis expansion (``Krylov_params``).
"""
import numpy as np

from ..linalg import np_conserved as npc
from ..linalg.krylov_based import LanczosEvolution
from ..linalg.truncation import svd_theta, TruncationError
from ..networks.mpo import MPOEnvironment
from .mps_common import OneSiteH, TwoSiteH, ZeroSiteH

__all__ = ['TDVPEngine', 'TwoSiteTDVPEngine', 'SingleSiteTDVPEngine']


class TDVPEngine:
    """Options: ``dt`` (0.1), ``N_steps`` (1), ``trunc_params``, ``lanczos_params`` (+ ``normalize``)."""

    def __init__(self, psi, model_H, options):
        if self.__class__ is TDVPEngine:
            raise NameError("use SingleSiteTDVPEngine or TwoSiteTDVPEngine")
        self.psi = psi
        self.H = model_H
        self.options = options = dict(options)
        self.trunc_params = dict(options.get('trunc_params', {}))
        self.lanczos_params = dict(options.get('lanczos_params', {}))
        self.env = MPOEnvironment(psi, model_H)
        self.dt = options.get('dt', 0.1)
        self.evolved_time = 0.
        self.trunc_err = TruncationError()
        self.trunc_err_list = []
        if not hasattr(psi, 'norm'):
            psi.norm = 1.

    def _krylov_evolve(self, H, theta, dt):
        return LanczosEvolution(H, theta, self.lanczos_params).run(dt, normalize=self.lanczos_params.get('normalize', None))

    def _site_changed(self, i):
        """Environments containing site i are stale."""
        env = self.env
        for j in range(i + 1, self.psi.L):
            env._LP[j] = None
        for j in range(0, i):
            env._RP[j] = None

    def sweep(self):
        self.trunc_err_list = []
        for i0, move_right, upd in self.get_sweep_schedule():
            self.update_local(i0, move_right, upd)
        return max(self.trunc_err_list) if self.trunc_err_list else 0.

    def evolve(self, N_steps, dt):
        """Evolve by ``N_steps * dt``; returns the accumulated truncation error (reference :211)."""
        self.dt = dt
        trunc_err = TruncationError()
        for _ in range(N_steps):
            self.sweep()
            for eps in self.trunc_err_list:
                trunc_err = trunc_err + TruncationError(eps, 1. - 2. * eps)
        self.evolved_time = self.evolved_time + N_steps * self.dt
        self.trunc_err = self.trunc_err + trunc_err
        return trunc_err

    def run(self):
        """``options['N_steps']`` steps of ``options['dt']`` (reference ``TimeEvolutionAlgorithm.run``)."""
        old_norm = self.psi.norm
        self.evolve(self.options.get('N_steps', 1), self.options.get('dt', 0.1))
        if self.options.get('preserve_norm', True):      # real time: keep the norm (reference ``run_evolution``)
            self.psi.norm = old_norm
        return self.psi


class TwoSiteTDVPEngine(TDVPEngine):
    def get_sweep_schedule(self):
        L = self.psi.L
        i0s = list(range(0, L - 2)) + list(range(L - 2, -1, -1))
        move_right = [True] * (L - 2) + [False] * (L - 2) + [None]
        upd = [[True, False]] * (L - 2) + [[False, True]] * (L - 2) + [[False, False]]
        return list(zip(i0s, move_right, upd))

    def update_local(self, i0, move_right, update_LP_RP):
        psi, L = self.psi, self.psi.L
        dt = -0.5j * self.dt
        if i0 == L - 2:
            dt = 2. * dt          # instead of updating the last pair of sites twice, double the time
        eff_H = TwoSiteH(self.env, i0, combine=True, move_right=move_right is not False)
        theta = eff_H.combine_theta(psi.get_theta(i0, n=2))
        theta, N = self._krylov_evolve(eff_H, theta, dt)
        theta = eff_H.prepare_svd(theta)
        qtotal_i0 = psi.get_B(i0, None).qtotal
        U, S, VH, err, renorm = svd_theta(theta, self.trunc_params, qtotal_LR=[qtotal_i0, None], inner_labels=['vR', 'vL'])
        psi.norm *= renorm
        psi.set_B(i0, U.split_legs(['(vL.p0)']).replace_label('p0', 'p'), form='A')
        psi.set_B(i0 + 1, VH.split_legs(['(p1.vR)']).replace_label('p1', 'p'), form='B')
        psi.set_SR(i0, S)
        self.trunc_err_list.append(err.eps)
        self._site_changed(i0)
        self._site_changed(i0 + 1)
        update_LP, update_RP = update_LP_RP
        if update_LP:
            eff_H.update_LP(self.env, i0 + 1, U)
        if update_RP:
            eff_H.update_RP(self.env, i0, VH)
        if move_right:
            self.one_site_update(i0 + 1, 0.5j * self.dt)
        elif move_right is False:
            self.one_site_update(i0, 0.5j * self.dt)
        return err

    def one_site_update(self, i, dt):
        """Backward evolution of the one-site wave function (reference :308)."""
        H1 = OneSiteH(self.env, i, combine=False)
        theta = H1.combine_theta(self.psi.get_theta(i, n=1))
        theta, _ = self._krylov_evolve(H1, theta, dt)
        self.psi.set_B(i, theta.replace_label('p0', 'p'), form='Th')
        self._site_changed(i)


class SingleSiteTDVPEngine(TDVPEngine):
    def get_sweep_schedule(self):
        L = self.psi.L
        i0s = list(range(0, L - 1)) + list(range(L - 1, -1, -1))
        move_right = [True] * (L - 1) + [False] * (L - 1) + [None]
        upd = [[True, False]] * (L - 1) + [[False, True]] * (L - 1) + [[False, False]]
        return list(zip(i0s, move_right, upd))

    def update_local(self, i0, move_right, update_LP_RP):
        L = self.psi.L
        dt = -0.5j * self.dt
        if i0 == L - 1:
            dt = 2. * dt
        eff_H = OneSiteH(self.env, i0, combine=True, move_right=bool(move_right))
        theta = eff_H.combine_theta(self.psi.get_theta(i0, n=1))
        theta, N = self._krylov_evolve(eff_H, theta, dt)
        if move_right:
            self.right_moving_update(eff_H, i0, theta)
        else:
            self.left_moving_update(eff_H, i0, theta)      # also the non-moving last update of a sweep
        self.trunc_err_list.append(0.)                      # no truncation in single-site TDVP

    def right_moving_update(self, eff_H, i0, theta):
        psi = self.psi
        U, S, VH = npc.svd(theta, qtotal_LR=[theta.qtotal, None], inner_labels=['vR', 'vL'])
        renorm = float(np.linalg.norm(S))
        S = S / renorm
        psi.norm *= renorm
        psi.set_B(i0, U.split_legs(['(vL.p0)']).replace_label('p0', 'p'), form='A')
        psi.set_SR(i0, S)
        self._site_changed(i0)
        eff_H.update_LP(self.env, i0 + 1, U.replace_label('(vL.p0)', '(vL.p)'))
        theta = VH.scale_axis(S, 'vL')
        theta = self.zero_site_update(i0 + 1, theta, 0.5j * self.dt)
        next_th = npc.tensordot(theta, psi.get_B(i0 + 1, 'B'), axes=['vR', 'vL'])
        psi.set_B(i0 + 1, next_th, form='Th')
        self._site_changed(i0 + 1)

    def left_moving_update(self, eff_H, i0, theta):
        psi = self.psi
        U, S, VH = npc.svd(theta, qtotal_LR=[None, theta.qtotal], inner_labels=['vR', 'vL'])
        renorm = float(np.linalg.norm(S))
        S = S / renorm
        psi.norm *= renorm
        if i0 == 0:
            assert U.shape == (1, 1)
            VH = VH * U.to_ndarray()[0, 0]       # just a global phase, but better keep it
        psi.set_B(i0, VH.split_legs(['(p0.vR)']).replace_label('p0', 'p'), form='B')
        psi.set_SL(i0, S)
        self._site_changed(i0)
        if i0 != 0:
            eff_H.update_RP(self.env, i0 - 1, VH.replace_label('(p0.vR)', '(p.vR)'))
            theta = U.scale_axis(S, 'vR')
            theta = self.zero_site_update(i0, theta, 0.5j * self.dt)
            next_th = npc.tensordot(psi.get_B(i0 - 1, 'A'), theta, axes=['vR', 'vL'])
            psi.set_B(i0 - 1, next_th, form='Th')
            self._site_changed(i0 - 1)
            # (the singular values changed by the zero-site step are deliberately NOT stored, reference :410-413)

    def zero_site_update(self, i, theta, dt):
        theta, _ = self._krylov_evolve(ZeroSiteH(self.env, i), theta, dt)
        return theta
