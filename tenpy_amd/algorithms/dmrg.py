"""Minimal two-site DMRG driver for boxes WITHOUT TeNPy (the GPU test box: ``bench.py``, ``smoke()``, the ``-m gpu`` tests).

Where TeNPy is installed, use TeNPy's own engines on the device mirror instead -- ``tenpy_amd.install.install()`` and then
``tenpy.algorithms.dmrg.TwoSiteDMRGEngine`` etc. run unchanged (mixers, single-site DMRG, infinite systems, excited states:
all of it is the reference's code, ``tests/test_reference_suite.py``).  This file only issues the npc calls of one bond update
in the reference's order (``prepare_update_local`` -> ``update_local`` (Lanczos, ``svd_theta``, ``set_B``) ->
``update_env``) for finite chains without a mixer, which is exactly BASELINE.json's timed workload.

Options (names as in the reference): ``trunc_params``, ``lanczos_params``, ``chi_list`` (dict sweep -> chi_max);
``shard_matvec`` (multi-GPU, ``algorithms/sharded.py``; with ``krylov_row_panels`` the Krylov vectors are row panels), ``profile`` (per-phase timers; synchronises the device).
"""
import pickle
import gc
import os
import time

import numpy as np

from ..linalg import np_conserved as npc
from ..linalg.krylov_based import LanczosGroundState
from ..linalg.truncation import svd_theta
from ..networks.mpo import MPOEnvironment
from .mps_common import TwoSiteH

__all__ = ['TwoSiteDMRGEngine']


_GC_PAUSE = os.environ.get('TPA_SWEEP_GC_PAUSE', '1') != '0'


class TwoSiteDMRGEngine:
    def _warm_token(self):
        """Key of this engine's warm-start bases (``linalg/_svd_warm.owner_token``: unique for the life of the engine, released with it)."""
        from ..linalg import _svd_warm
        return _svd_warm.owner_token(self)

    def __init__(self, psi, model_H, options, resume_data=None):
        if not psi.finite:
            raise ValueError("the stand-alone driver handles finite chains; run TeNPy's engines on the mirror for the rest")
        self.psi, self.H = psi, model_H
        self.options = options = dict(options)
        self.trunc_params = dict(options.get('trunc_params', {}))
        self.lanczos_params = dict(options.get('lanczos_params', {}))
        self.chi_list = options.get('chi_list')
        self.env = MPOEnvironment(psi, model_H)
        self.sweeps = 0
        self.update_stats = {k: [] for k in ('i0', 'E_total', 'N_lanczos', 'time', 'err', 'chi')}
        self.sweep_stats = {k: [] for k in ('sweep', 'E', 'S', 'time', 'max_trunc_err', 'max_chi')}
        self.profile = bool(options.get('profile', False))
        self.phase_time = {'heff': 0., 'lanczos': 0., 'svd': 0., 'env': 0., 'setB': 0.}
        self.shard_matvec = bool(options.get('shard_matvec', False))
        self._svd_group = None
        if self.shard_matvec:           # multi-GPU: the independent charge blocks of every SVD are dealt out to the ranks too
            import torch.distributed as dist
            import os
            if dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get('TPA_SHARD_FORCE')):
                self._svd_group = (None, dist.get_rank(), dist.get_world_size())
        if resume_data is not None:
            self.sweeps = int(resume_data['sweeps'])
            for name in ('sweep_stats', 'update_stats'):
                for k, v in resume_data.get(name, {}).items():
                    getattr(self, name)[k] = list(v)
            if resume_data.get('chi_max') is not None:
                self.trunc_params['chi_max'] = resume_data['chi_max']
            if resume_data.get('lanczos_params') is not None:
                self.lanczos_params = dict(resume_data['lanczos_params'])

    # ---- checkpoint / resume: the state (device arrays pickle through the host), counters, statistics --------------------
    def get_resume_data(self):
        """Not interchangeable with the reference's checkpoints (those hold TeNPy's own MPS class); environments are rebuilt
        from the state on demand."""
        return {'psi': self.psi, 'sweeps': self.sweeps, 'chi_max': self.trunc_params.get('chi_max'),
                'lanczos_params': dict(self.lanczos_params),
                'sweep_stats': {k: list(v) for k, v in self.sweep_stats.items()},
                'update_stats': {k: list(v) for k, v in self.update_stats.items()}}

    def save_checkpoint(self, filename):
        with open(filename, 'wb') as f:
            pickle.dump(self.get_resume_data(), f, protocol=4)

    @classmethod
    def from_checkpoint(cls, filename, model_H, options):
        with open(filename, 'rb') as f:
            data = pickle.load(f)
        return cls(data['psi'], model_H, options, resume_data=data)

    # ---- one sweep = 2 (L - 2) bond updates ----------------------------------------------------------------------------------
    def sweep(self):
        if self.chi_list is not None:
            due = [s for s in self.chi_list if s <= self.sweeps]
            if due:
                self.trunc_params['chi_max'] = self.chi_list[max(due)]
        L = self.psi.L
        t0 = time.time()
        worst = 0.
        # The cyclic garbage collector is paused for the duration of a sweep: a bond update allocates ~2000 short-lived containers
        # (all freed by reference counting), which triggers a generation-0 pass every few hundred microseconds and, every ~100 bonds, a
        # full pass over the plan caches (10^5 objects) -- host time in front of an idle device.  Collected once per sweep instead.
        gc_was_on = _GC_PAUSE and gc.isenabled()
        if gc_was_on:
            gc.disable()
        try:
            for i0 in range(L - 2):
                worst = max(worst, self.update_bond(i0, move_right=True).eps)
            for i0 in range(L - 2, 0, -1):
                worst = max(worst, self.update_bond(i0, move_right=False).eps)
        finally:
            if gc_was_on:
                gc.enable()
        self.sweeps += 1
        st = self.sweep_stats
        st['sweep'].append(self.sweeps)
        st['E'].append(self.update_stats['E_total'][-1])
        st['S'].append(float(np.max(self.psi.entanglement_entropy())))
        st['time'].append(time.time() - t0)
        st['max_trunc_err'].append(worst)
        st['max_chi'].append(int(np.max(self.psi.chi)))
        return worst

    def update_bond(self, i0, move_right=True):
        t0 = time.time()
        psi, env = self.psi, self.env
        self._tick(None)
        if self.shard_matvec:
            from .sharded import ShardedTwoSiteH
            eff_H = ShardedTwoSiteH(env, i0, combine=True, move_right=move_right)
            # the row-panel tables of this bond's previous visit (validated by their keys in `matvec_program`): building them is
            # ~0.8 ms of host time per bond, with the device idle
            eff_H._sharded_cache = self.__dict__.setdefault('_sharded_plans', {}).setdefault(i0, {})
        else:
            eff_H = TwoSiteH(env, i0, combine=True, move_right=move_right)
            # the contraction plans of this bond's previous visit (same block structure once chi has saturated: checked by their keys)
            cache = self.__dict__.setdefault('_heff_plans', {})
            if eff_H.factored and cache.get(i0) is not None:
                eff_H._fplans = cache[i0]
        theta = eff_H.combine_theta(psi.get_theta(i0, n=2))
        self._tick('heff')
        if self.shard_matvec and self.options.get('krylov_row_panels', False):
            from .sharded import lanczos_row_panels          # north_star: Krylov vectors as row panels, scalars by all-reduce
            E0, theta, N = lanczos_row_panels(eff_H, theta, self.lanczos_params)
        else:
            E0, theta, N = LanczosGroundState(eff_H, theta, self.lanczos_params).run()
        if not self.shard_matvec and isinstance(getattr(eff_H, '_fplans', None), dict) and 'lkey' in eff_H._fplans:
            self._heff_plans[i0] = eff_H._fplans
        theta = eff_H.prepare_svd(theta)                      # fused matrix [(vL.p0), (p1.vR)]
        self._tick('lanczos')
        previous = npc.SVD_DIST_GROUP
        npc.SVD_DIST_GROUP = self._svd_group                  # scoped to this call (other SVDs in the process stay local)
        # warm start of the block SVD: the right (left) singular vectors this bond produced on the previous visit span
        # theta_0 = M . B_{i0+1} (A_{i0} . M) of this one (linalg/_svd_warm.py); consumed by the next npc.svd call
        npc.svd_hint = ((self._warm_token(), i0), 'R' if move_right else 'L')
        try:
            U, S, VH, err, _ = svd_theta(theta, self.trunc_params, qtotal_LR=[psi.get_B(i0, None).qtotal, None],
                                         inner_labels=['vR', 'vL'])
        finally:
            npc.SVD_DIST_GROUP = previous
            npc.svd_hint = None
        self._tick('svd')
        i1 = i0 + 1
        if move_right:
            eff_H.update_LP(env, i1, U)
        else:
            eff_H.update_RP(env, i0, VH)
        self._tick('env')
        psi.set_B(i0, U.split_legs(['(vL.p0)']).ireplace_label('p0', 'p'), form='A')
        psi.set_B(i1, VH.split_legs(['(p1.vR)']).ireplace_label('p1', 'p'), form='B')
        psi.set_SR(i0, S)
        env.invalidate(i0, i1, keep_LP=move_right, keep_RP=not move_right)
        self._tick('setB')
        us = self.update_stats
        us['i0'].append(i0)
        us['E_total'].append(float(E0))
        us['N_lanczos'].append(N)
        us['time'].append(time.time() - t0)
        us['err'].append(err.eps)
        us['chi'].append(len(S))
        return err

    def _tick(self, phase):
        """Phase timer (the reference's DEBUG_PRINT phases, _npc_helper.pyx:13); synchronises only when profiling."""
        if not self.profile:
            return
        from ..linalg import _device as dev
        dev.torch().cuda.synchronize()
        now = time.time()
        if phase is not None:
            self.phase_time[phase] += now - self._t_phase
        self._t_phase = now
