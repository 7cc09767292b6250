"""Lanczos ground-state search on device-resident block vectors.

Mirrors ``tenpy/linalg/krylov_based.py`` (``KrylovBased`` :28, ``LanczosGroundState`` :584,
``_build_krylov`` :645, ``_converged`` :678, ``_calc_result_full`` :160) -- same options, same
convergence criteria, same returned triple ``(E0, psi0, N)`` -- but the three-term recurrence is fused:

* all Krylov vectors of one run share ONE block structure, so every BLAS-1 step is a flat pass over
  the packed arenas;
* ``w -= alpha v_k ; w -= beta v_{k-1} ; |w|^2`` is a single kernel (``tpa_lanczos_update``) instead of
  two axpy + a norm (reference :665-672); alpha and beta are the only two host synchronisations per
  iteration (the tridiagonal ``eigh`` stays on the host).
"""
import logging

import numpy as np

from . import _device as dev
from . import np_conserved as npc

logger = logging.getLogger(__name__)

import os as _os
NATIVE = _os.environ.get('TPA_LANCZOS_NATIVE', '1') != '0'       # the whole run as one C-ABI call (tpa_lanczos_run) where the operator offers a launch program
PIPELINED = _os.environ.get('TPA_LANCZOS_PIPELINED', '1') != '0'     # device-resident alpha / beta in LanczosGroundState (off: one host read per scalar, as in round 1)

__all__ = ['LanczosGroundState', 'LanczosEvolution', 'Arnoldi', 'lanczos', 'gram_schmidt', 'iscale_prefactor', 'iadd_prefactor_other']


stats = {'runs': 0, 'n_matvec': 0, 'n_ill_conditioned': 0, 'n_degenerate': 0, 'n_native_sharded': 0}


class LanczosGroundState:
    """Lanczos algorithm for the ground state of a hermitian ``H`` given through ``H.matvec(vec)``.

    Options (defaults as in the reference): ``N_min`` 2, ``N_max`` 20, ``P_tol`` 1e-14, ``min_gap`` 1e-12,
    ``E_tol`` inf, ``N_cache`` N_max, ``cutoff`` 100*eps, ``reortho`` False, ``E_shift`` None.
    """

    def __init__(self, H, psi0, options):
        self.H = H
        self.psi0 = psi0.copy(deep=True)
        self.options = options = dict(options) if options is not None else {}
        self.N_min = int(options.get('N_min', 2))
        self.N_max = int(options.get('N_max', 20))
        self.P_tol = options.get('P_tol', 1.e-14)
        self.min_gap = options.get('min_gap', 1.e-12)
        self.reortho = bool(options.get('reortho', False))
        self.E_shift = options.get('E_shift', None)
        self.E_tol = options.get('E_tol', np.inf)
        self.N_cache = int(options.get('N_cache', self.N_max))
        if self.N_min < 2:
            raise ValueError("Should perform at least 2 steps.")
        if self.N_cache < 2:
            raise ValueError("Need to cache at least two vectors.")
        self._cutoff = options.get('cutoff', np.finfo(np.float64).eps * 100)
        self._cache = []
        self._psi0_norm = None
        self.Es = np.zeros([self.N_max, self.N_max], dtype=np.float64)
        self._h_krylov = np.zeros([self.N_max + 1, self.N_max + 1], dtype=np.float64)
        self._result_krylov = None

    # ---- public -----------------------------------------------------------------------------------------
    def run(self):
        """Returns ``(E0, psi0, N)``: energy estimate, normalised ground state estimate, iterations."""
        stats['runs'] += 1
        prog = self._native_program()
        if prog is not None:
            return self._run_native(prog)
        N = self._build_krylov()
        E0 = self.Es[N - 1, 0]
        if self.E_shift is not None:
            E0 -= self.E_shift
        if N == 1:
            return E0, self.psi0.copy(deep=True), N
        return E0, self._calc_result_full(N), N

    # ---- the whole run as one host call (tpa_lanczos_run) ---------------------------------------------------------
    def _native_program(self):
        """``(ops, bufs, gemm_plans)`` if the operator can hand its matvec over as a launch program (``TwoSiteH.matvec_program``)
        and the options are the ones the native loop covers (no re-orthogonalisation, all Krylov vectors cached)."""
        if not (NATIVE and PIPELINED) or self.reortho or self.N_cache < self.N_max or self.N_max + 1 > 64:
            return None
        make = getattr(self.H, 'native_input', None)
        if make is None:
            return None
        w = self.psi0
        if w.stored_blocks == 0 or not w._is_packed():
            return None
        got = make(w)
        if got is None:
            return None
        self.psi0, prog = got          # (possibly theta embedded in the block structure of H theta)
        return prog

    def _run_native(self, prog):
        """``_build_krylov`` + ``_calc_result_full`` through ``tpa_lanczos_run`` / ``tpa_krylov_combine``: the device side of every
        step is enqueued by a C++ loop, the reference's host side of a step (tridiagonal ``eigh``, ``_converged``) runs in the
        callback one step late -- the same numbers as :meth:`_build_krylov_pipelined`, without ~0.5 ms of interpreter time per step.

        Sharded operators (``collective`` callback): the failure agreement at the end only covers errors raised AFTER the run's last
        collective.  A rank that fails mid-run (HIP error, or the collective callback returning 1) reaches the agreement all-reduce while
        the others sit in the next all-gather of the same communicator -- mismatched collectives, which hang until the process group's
        timeout instead of raising the intended error (set a finite ``timeout`` on ``init_process_group``).  The success path
        pays one extra all-reduce and one host read per bond for the agreement."""
        from .. import _lib
        ops, bufs, gemm_plans = prog[:3]
        collective = prog[3] if len(prog) > 3 else None       # sharded operators: the all-gather of the row panels (op kind 3)
        w = self.psi0
        n, dtype = w._arena.numel(), w.dtype
        code, L = dev.code(dtype), dev.lib()
        # The launch program exists only when the block structure of the vector is CLOSED under the operator (`native_input`): the
        # Krylov space then lives in n dimensions and is exhausted after n steps -- the exact answer in that space.  A forced N_min
        # beyond n would only rotate rounding noise (edge bonds of a chain: n = 3 ... 6; beta stays far above any cutoff when the
        # start vector was nearly converged, because its tiny first residual is normalised).  The reference's engine never gets there:
        # it diagonalises effective Hamiltonians below N = 400 exactly (algorithms/dmrg.py:733-739, `full_diag_effH`).
        N_max = max(2, min(self.N_max, n))
        N_min_run = min(self.N_min, N_max)
        krylov = dev.scratch('lanczos_krylov', (N_max + 1) * n, dtype)
        scal = dev.scratch('lanczos_scalars', 2 * (N_max + 2) + 4, np.float64)
        _, scr = dev.reduction_buffers()
        h = self._h_krylov
        err = []

        def record(j, alpha, bsq, user):
            try:
                b = float(np.sqrt(bsq))
                h[j, j] = alpha
                # the tridiagonal eigen-problem of step j is only looked at by the stopping test of steps j and j + 1 (reference
                # :673, `_converged` reads Es[j] and Es[j - 1]) and by the final result: below N_min - 2 it is skipped -- the same
                # (E0, psi0, N), ~30 us less host time per step (at chi <= 512 the device waits for this callback)
                # (also at the last step the loop can take: the reference allows N_min > N_max and returns the N_max result)
                if j + 2 >= N_min_run or j + 1 >= N_max or abs(b) < self._cutoff:
                    self._calc_result_krylov(j)
                h[j, j + 1] = h[j + 1, j] = b
                return int(abs(b) < self._cutoff or (j + 1 >= N_min_run and self._converged(j)))
            except BaseException as e:       # an exception must not unwind through the C frame
                err.append(e)
                return 1
        cb = _lib.LANCZOS_CALLBACK(record)
        ptrs = np.array([t.data_ptr() for t in bufs], dtype=np.int64)
        info = np.zeros(4, dtype=np.float64)
        timed = npc.gemm_timer.sample()
        ccb = None
        if collective is not None:
            ccb = _lib.COLLECTIVE_CALLBACK(collective)
            L.tpa_lanczos_set_collective(ccb, None)
            stats['n_native_sharded'] = stats.get('n_native_sharded', 0) + 1
        failure = None
        try:
            dev.check(L.tpa_lanczos_run(code, n, ops.ctypes.data, len(ops), ptrs.ctypes.data, len(ptrs), krylov.data_ptr(),
                                        w._arena.data_ptr(), N_max, float(self._cutoff), int(self.E_shift is not None),
                                        float(self.E_shift or 0.), scal.data_ptr(), scr.data_ptr(), cb, None, int(timed),
                                        info.ctypes.data, dev.stream()), "lanczos_run")
        except Exception as e:
            if collective is None:
                raise
            failure = e
        finally:
            if ccb is not None:
                L.tpa_lanczos_set_collective(_lib.COLLECTIVE_CALLBACK(), None)
        if err and failure is None:
            failure = err[0]
        agree = getattr(collective, 'agree', None)
        if agree is not None:
            # sharded operators: a rank that raised alone would leave the others waiting in the NEXT collective (the SVD gather of
            # the same bond) forever -- agree on the outcome of the run before anybody goes on
            if agree(failure is not None) and failure is None:
                failure = RuntimeError("tenpy_amd Lanczos: the run failed on another rank")
        if failure is not None:
            raise failure
        N, n_mv = int(info[0]), int(info[1])
        stats['n_matvec'] += n_mv
        if timed:
            gt = npc.gemm_timer
            gt.n_launch += n_mv * len(gemm_plans)
            gt.flops += n_mv * sum(p.flops for p in gemm_plans)
            gt.bytes_min += n_mv * sum(p.bytes_min for p in gemm_plans)
            gt.ms += float(info[2])
        if N == 0:
            raise ValueError("Norm of self.psi0 too small: {0}".format(info[3]))
        if self._psi0_norm is None:
            self._psi0_norm = float(info[3])
        E0 = self.Es[N - 1, 0]
        if self.E_shift is not None:
            E0 -= self.E_shift
        psif = w.copy(deep=False)
        out = dev.empty(n, dtype)
        if N == 1:
            out.copy_(krylov[:n])
            psif._arena = out
            return E0, psif, N
        vf = np.ascontiguousarray(self._result_krylov, dtype=np.float64)
        assert len(vf) == N
        nrm = np.zeros(1, dtype=np.float64)
        dev.check(L.tpa_krylov_combine(code, n, krylov.data_ptr(), N, vf.ctypes.data, out.data_ptr(), scal.data_ptr() + 8 * 2 * (N_max + 2),
                                       scr.data_ptr(), nrm.ctypes.data, dev.stream()), "krylov_combine")
        nrm = float(nrm[0])
        psif._arena = out
        if abs(1. - nrm) > 1.e-5:
            stats['n_ill_conditioned'] += 1
            logger.warning("poorly conditioned H matrix in KrylovBased! |psi_0| = %f", nrm)
        if not (nrm > 1.e-8) or not np.isfinite(nrm):      # see _calc_result_full
            stats['n_degenerate'] += 1
            out.copy_(krylov[:n])
            nrm = 1.
        if nrm != 1.:
            dev.check(L.tpa_scal(code, n, 1. / nrm, 0., out.data_ptr(), dev.stream()), "scal")
        return E0, psif, N

    # ---- internals ------------------------------------------------------------------------------------------
    def _matvec(self, w):
        stats['n_matvec'] += 1
        r = self.H.matvec(w)
        if self.E_shift is not None:
            r.iadd_prefactor_other(self.E_shift, w)
        return r

    def _flat_ok(self, w, *others):
        return w.stored_blocks > 0 and w._is_packed() and all(w._same_structure(o) and w.dtype == o.dtype for o in others)

    def _build_krylov(self):
        w = self.psi0
        beta = npc.norm(w)
        if beta < self._cutoff:
            raise ValueError("Norm of self.psi0 too small: {0}".format(beta))
        if self._psi0_norm is None:
            self._psi0_norm = beta
        if not self.reortho and PIPELINED and w.stored_blocks > 0 and w._is_packed():
            return self._build_krylov_pipelined(w, beta)
        return self._build_krylov_stepwise(w, beta)

    def _build_krylov_pipelined(self, w, beta):
        """The recurrence with alpha / beta kept on the device (``tpa_lanczos_step``): per step the host enqueues matvec,
        dot, update, norm and normalisation without waiting, and reads the two scalars of the PREVIOUS step for the
        tridiagonal eigen-problem and the stopping test of the reference (:673) while the device is already working on
        the next matvec.  If that test says "stop at k", the step in flight is discarded -- the same Krylov space, the same
        (E0, psi0, N) as the step-by-step loop."""
        h = self._h_krylov
        L = dev.lib()
        pipe = dev.ScalarPipe(self.N_max + 1)
        _, scr = dev.reduction_buffers()
        code = dev.code(w.dtype)
        w.iscale_prefactor(1. / beta)

        done = [False] * (self.N_max + 1)

        def record(j, alpha, b):          # host side of step j; True = stop after it
            done[j] = True
            h[j, j] = alpha
            self._calc_result_krylov(j)
            h[j, j + 1] = h[j + 1, j] = b
            return abs(b) < self._cutoff or (j + 1 >= self.N_min and self._converged(j))

        def finish(j):
            alpha, bsq = pipe.get(j)
            return record(j, alpha, float(np.sqrt(bsq)))
        k = 0
        for k in range(self.N_max):
            self._to_cache(w)
            v1 = self._cache[-1]
            w = self._matvec(v1)
            v0 = self._cache[-2] if k > 0 else None
            if self._flat_ok(w, v1, *([v0] if v0 is not None else [])):
                w._own_arena()          # copy-on-write: a matvec may hand back a shallow copy of its input
                dev.check(L.tpa_lanczos_step(code, w._arena.numel(), w._arena.data_ptr(), v1._arena.data_ptr(),
                                             v0._arena.data_ptr() if v0 is not None else None,
                                             pipe.ptr(k - 1, 1) if v0 is not None else None, pipe.ptr(k), scr.data_ptr(),
                                             dev.stream()), "lanczos_step")
                pipe.post(k)
                if k > 0 and not done[k - 1] and finish(k - 1):
                    self._cache.pop()       # v_k was cached for a step that is not part of the result
                    return k
                continue
            # the operator created / dropped blocks (e.g. the first steps from a product state): this step the slow way
            if k > 0 and not done[k - 1] and finish(k - 1):
                self._cache.pop()
                return k
            alpha = float(np.real(npc.inner(w, v1, axes='range', do_conj=True)))
            w.iadd_prefactor_other(-alpha, v1)
            if v0 is not None:
                w.iadd_prefactor_other(-h[k - 1, k], v0)
            b = npc.norm(w)
            pipe.dev[k, 0], pipe.dev[k, 1] = alpha, b * b        # the next (device-side) step reads beta from here
            stop = record(k, alpha, b)
            if stop:
                return k + 1
            w.iscale_prefactor(1. / b)
            if not w._is_packed():
                w._repack()
        if not done[k]:
            finish(k)
        return k + 1

    def _build_krylov_stepwise(self, w, beta):
        h = self._h_krylov
        k = 0
        for k in range(self.N_max):
            w.iscale_prefactor(1. / beta)
            self._to_cache(w)
            w = self._matvec(w)
            v1 = self._cache[-1]
            alpha = float(np.real(npc.inner(w, v1, axes='range', do_conj=True)))
            h[k, k] = alpha
            self._calc_result_krylov(k)
            v0 = self._cache[-2] if (k > 0 and not self.reortho) else None
            beta_prev = beta
            if self._flat_ok(w, v1, *([v0] if v0 is not None else [])):
                out, scr = dev.reduction_buffers()
                w._own_arena()
                dev.check(dev.lib().tpa_lanczos_update(
                    dev.code(w.dtype), w._arena.numel(), w._arena.data_ptr(), alpha, 0., v1._arena.data_ptr(),
                    beta_prev, 0., v0._arena.data_ptr() if v0 is not None else None, out.data_ptr(), scr.data_ptr(),
                    dev.stream()), "lanczos_update")
                if self.reortho:
                    for c in self._cache[:-1]:
                        w.iadd_prefactor_other(-npc.inner(c, w, axes='range', do_conj=True), c)
                    beta = npc.norm(w)
                else:
                    beta = float(np.sqrt(dev.read_scalar(out, False)))
            else:  # generic path (block structures differ, e.g. H creates / drops blocks)
                w.iadd_prefactor_other(-alpha, v1)
                if self.reortho:
                    for c in self._cache[:-1]:
                        w.iadd_prefactor_other(-npc.inner(c, w, axes='range', do_conj=True), c)
                elif k > 0:
                    w.iadd_prefactor_other(-beta_prev, self._cache[-2])
                beta = npc.norm(w)
            h[k, k + 1] = h[k + 1, k] = beta
            if abs(beta) < self._cutoff or (k + 1 >= self.N_min and self._converged(k)):
                break
        return k + 1

    def _converged(self, k):
        v0 = self._result_krylov
        E = self.Es[k, :]
        ritz_res = abs(v0[k]) * self._h_krylov[k, k + 1]
        gap = max(E[1] - E[0], self.min_gap)
        p_err = (ritz_res / gap)**2
        delta_E0 = self.Es[k - 1, 0] - E[0]
        return p_err < self.P_tol and delta_E0 < self.E_tol

    def _calc_result_krylov(self, k):
        h = self._h_krylov
        if k == 0:
            self.Es[0, 0] = h[0, 0]
            self._result_krylov = np.ones(1, np.float64)
        else:
            E_kr, v_kr = np.linalg.eigh(h[:k + 1, :k + 1])
            self.Es[k, :k + 1] = E_kr
            self._result_krylov = v_kr[:, 0]

    def _to_cache(self, psi):
        self._cache.append(psi)
        if len(self._cache) > self.N_cache:
            self._cache.pop(0)

    def _calc_result_full(self, N):
        """``psi = sum_k c_k v_k`` over the Krylov ONB; vectors that fell out of the cache are re-generated."""
        vf = self._result_krylov
        assert N == len(vf) > 1
        psif = self.psi0 * vf[0]
        len_cache = len(self._cache)
        terms = [(vf[N - k], self._cache[-k]) for k in range(1, min(len_cache + 1, N))]
        key = psif._struct_key()
        if psif.stored_blocks and psif._is_packed() and all(
                v.dtype == psif.dtype and v._struct_key() == key and all(a is b for a, b in zip(v.legs, psif.legs))
                for _, v in terms):
            # the Krylov vectors of one run share block structure and leg objects: flat axpys on the arenas, without the
            # per-call argument checks of iadd_prefactor_other (11 of them per bond update cost 0.7 ms of host time)
            L, code, n, st = dev.lib(), dev.code(psif.dtype), psif._arena.numel(), dev.stream()
            psif._own_arena()
            for c, v in terms:
                c = complex(c)
                dev.check(L.tpa_axpy(code, n, c.real, c.imag, v._arena.data_ptr(), psif._arena.data_ptr(), st), "axpy")
        else:
            for c, v in terms:
                psif.iadd_prefactor_other(c, v)
        self._cache = []
        self._rebuild_krylov_for_result_full(psif, N - len_cache - 1)
        nrm = npc.norm(psif)
        if abs(1. - nrm) > 1.e-5:
            stats['n_ill_conditioned'] += 1
            logger.warning("poorly conditioned H matrix in KrylovBased! |psi_0| = %f", nrm)
        if not (nrm > 1.e-8) or not np.isfinite(nrm):
            # The Krylov vectors of a run that was forced past convergence (N_min beyond the point where beta hits rounding
            # level) are noise, and their combination can cancel to (numerically) nothing; the reference divides by that norm
            # (krylov_based.py:236-239) and hands NaNs to the SVD.  The start vector is the best available answer then.
            stats['n_degenerate'] += 1
            psif = self.psi0.copy(deep=True)
            nrm = npc.norm(psif)
        psif.iscale_prefactor(1. / nrm)
        return psif

    def _rebuild_krylov_for_result_full(self, psif, N_max):
        vf, h = self._result_krylov, self._h_krylov
        w = self.psi0
        beta = None
        for k in range(0, N_max):
            self._to_cache(w)
            w = self._matvec(w)
            w.iadd_prefactor_other(-h[k, k], self._cache[-1])
            if self.reortho:
                for c in self._cache[:-1]:
                    w.iadd_prefactor_other(-npc.inner(c, w, axes='range', do_conj=True), c)
            elif k > 0:
                w.iadd_prefactor_other(-beta, self._cache[-2])
            beta = h[k, k + 1]
            w.iscale_prefactor(1. / beta)
            psif.iadd_prefactor_other(vf[k + 1], w)


class LanczosEvolution(LanczosGroundState):
    """``exp(delta H) |psi0>`` from the same Krylov recurrence (reference :718-822): instead of the ground state of the
    tridiagonal ``h`` take ``exp(delta h) e_0``; converged when the last Krylov component drops below ``P_tol``.
    The device work is identical to the ground-state search (matvec + the fused vector update per iteration)."""

    def __init__(self, H, psi0, options):
        super().__init__(H, psi0, options)
        self._result_norm = 1.
        self.delta = None

    def run(self, delta, normalize=None):
        """Returns ``(psi_f, N)``; ``normalize`` defaults to ``real(delta) == 0`` (unitary evolution)."""
        self.delta = delta
        N = self._build_krylov()
        if N == 1:
            result_full = self.psi0 * self._result_krylov[0]      # only a phase
        else:
            result_full = self._calc_result_full(N)
        if normalize is None:
            normalize = np.real(delta) == 0.
        if normalize:
            return result_full, N
        return result_full * (self._psi0_norm * self._result_norm), N

    def _calc_result_krylov(self, k):
        h, delta = self._h_krylov, self.delta
        if k == 0:
            exp_dE = np.exp(delta * h[0, 0])
            self._result_norm = np.abs(exp_dE)
            self._result_krylov = np.array([exp_dE / self._result_norm])
        else:
            E_kr, v_kr = np.linalg.eigh(h[:k + 1, :k + 1])
            exp_dH_e0 = np.dot(v_kr, np.exp(E_kr * delta) * np.conj(v_kr[0, :]))
            self._result_norm = np.linalg.norm(exp_dH_e0)
            self._result_krylov = exp_dH_e0 / self._result_norm

    def _converged(self, k):
        return np.abs(self._result_krylov[k]) < self.P_tol


class Arnoldi:
    """Arnoldi iteration for the dominant eigenvector(s) of a general (non-Hermitian) operator given through
    ``H.matvec`` -- e.g. the transfer matrix of an infinite MPS (reference ``krylov_based.Arnoldi`` :322).

    Options: ``N_min`` (2), ``N_max`` (20), ``P_tol`` (1e-14), ``E_tol`` (inf), ``min_gap`` (1e-12), ``cutoff``, ``which`` ('LM':
    largest magnitude, 'LR' / 'SR': largest / smallest real part), ``num_ev`` (1).  The Krylov basis stays on the device
    (full Gram-Schmidt: k inner products + k axpy in step k); the Hessenberg matrix is diagonalised on the host."""

    def __init__(self, H, psi0, options):
        self.H = H
        self.psi0 = psi0.copy(deep=True)
        opt = dict(options) if options is not None else {}
        self.N_min = int(opt.get('N_min', 2))
        self.N_max = int(opt.get('N_max', 20))
        self.P_tol = opt.get('P_tol', 1.e-14)
        self.E_tol = opt.get('E_tol', np.inf)
        self.min_gap = opt.get('min_gap', 1.e-12)
        self.which = opt.get('which', 'LM')
        self.num_ev = int(opt.get('num_ev', 1))
        self._cutoff = opt.get('cutoff', np.finfo(np.float64).eps * 100)
        self._basis = []
        self._hess = np.zeros((self.N_max + 1, self.N_max), dtype=np.complex128)
        self.Es = np.zeros((self.N_max, self.N_max), dtype=np.complex128)
        self._ritz = None

    def _order(self, ev):
        key = {'LM': -np.abs(ev), 'LR': -np.real(ev), 'SR': np.real(ev), 'SM': np.abs(ev)}[self.which]
        return np.argsort(key, kind='stable')

    def run(self):
        """Returns ``(eigenvalues[num_ev], eigenvectors[<= num_ev], N)``."""
        hs = self._hess
        w = self.psi0
        nrm = npc.norm(w)
        N = 0
        for k in range(self.N_max):
            w.iscale_prefactor(1. / nrm)
            self._basis.append(w)
            w = self.H.matvec(w)
            for i, v in enumerate(self._basis):
                ov = npc.inner(v, w, axes='range', do_conj=True)
                hs[i, k] = ov
                w.iadd_prefactor_other(-ov, v)
            hs[k + 1, k] = nrm = npc.norm(w)
            N = k + 1
            if k == 0:
                self.Es[0, 0] = hs[0, 0]
                self._ritz = np.ones((1, 1), dtype=np.complex128)
            else:
                ev, vec = np.linalg.eig(hs[:k + 1, :k + 1])
                order = self._order(ev)
                self.Es[k, :k + 1] = ev[order]
                self._ritz = vec[:, order]
            if nrm < self._cutoff or (N >= self.N_min and self._converged(k)):
                break
        E = self.Es[N - 1, :self.num_ev].copy()
        if N == 1:
            return np.real_if_close(E), [self.psi0.copy(deep=True)], N
        out = []
        for j in range(min(N, self.num_ev)):
            c = np.real_if_close(self._ritz[:, j])
            psi = self._basis[0] * c[0]
            for k in range(1, N):
                psi.iadd_prefactor_other(c[k], self._basis[k])
            psi.iscale_prefactor(1. / npc.norm(psi))
            out.append(psi)
        return np.real_if_close(E), out, N

    def _converged(self, k):
        if k == 0:
            return False
        v0 = self._ritz[:, 0]
        E = self.Es[k, :k + 1]
        ritz_res = abs(v0[k]) * abs(self._hess[k + 1, k])
        gaps = [np.min(np.abs(E[i + 1:] - E[i])) for i in range(min(self.num_ev, k))]
        gap = max(min(gaps) if gaps else self.min_gap, self.min_gap)
        p_err = (ritz_res / gap)**2
        delta_E = abs(self.Es[k - 1, 0] - E[0])
        return p_err < self.P_tol and delta_E < self.E_tol


def gram_schmidt(vecs, rcond=1.e-14):
    """In-place Gram-Schmidt ortho-normalisation of a list of Arrays with the same leg order (reference :858); vectors
    whose remainder has norm <= ``rcond`` are dropped."""
    res = []
    for vec in vecs:
        for other in res:
            ov = npc.inner(other, vec, axes='range', do_conj=True)
            iadd_prefactor_other(vec, -ov, other)
        n = npc.norm(vec)
        if n > rcond:
            iscale_prefactor(vec, 1. / n)
            res.append(vec)
    return res


def iscale_prefactor(w, scale):
    """``w *= scale`` for an Array or a list of Arrays (reference :888)."""
    if not isinstance(w, list):
        w.iscale_prefactor(scale)
    else:
        for a in w:
            a.iscale_prefactor(scale)


def iadd_prefactor_other(w, alpha, v):
    """``w += alpha v`` for Arrays or lists of Arrays (reference :896)."""
    if not isinstance(w, list):
        w.iadd_prefactor_other(alpha, v)
    else:
        for a, b in zip(w, v):
            a.iadd_prefactor_other(alpha, b)


def lanczos(H, psi, options={}, orthogonal_to=[]):
    """Convenience wrapper like the reference's (deprecated) ``lanczos`` function."""
    if len(orthogonal_to):
        from .sparse import OrthogonalNpcLinearOperator
        H = OrthogonalNpcLinearOperator(H, [v.copy(deep=True) for v in orthogonal_to])
    return LanczosGroundState(H, psi, options).run()
