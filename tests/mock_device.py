"""TEST INFRASTRUCTURE ONLY: a numpy emulation of the *device* entry points of the C-ABI.

Purpose: exercise the HOST logic of ``tenpy_amd`` (charge bookkeeping, contraction/copy plans, leg
fusion, truncation, Lanczos driver ...) in ``-m "not gpu"`` tests in a container without a GPU.  The
product never imports this file; without the ``mock_device`` fixture every compute call raises
``BackendError`` on a GPU-less machine (``tests/test_no_fallback.py`` checks exactly that).

The emulation follows the *documented contract* of ``include/tenpy_amd.h`` (table layouts, strides,
job formats), so it also pins that contract: a host-side change that builds wrong tables fails here.
Host-only entry points (``tpa_plan_tensordot``, ``tpa_gemm_tile_shape``, ...) are forwarded to the real
shared library.
"""
import bisect
import threading
import ctypes
import weakref

import numpy as np
import scipy.linalg
import torch

from tenpy_amd import _lib
from tenpy_amd.linalg import _device as dev

MAXD = 6


class _Registry:
    """Maps raw addresses back to the (still alive) CPU tensors that own them."""

    def __init__(self):
        self.starts, self.refs = [], []
        self.n_add = 0
        self.lock = threading.RLock()       # the reference's `+ h.c.` worker (dmrg_parallel.py) contracts in a second thread

    def _purge(self):
        alive = [(s, r) for s, r in zip(self.starts, self.refs) if r() is not None]
        self.starts = [s for s, _ in alive]
        self.refs = [r for _, r in alive]

    def add(self, t):
        if t.numel() == 0:
            return t
        with self.lock:
            return self._add(t)

    def _add(self, t):
        self.n_add += 1
        if self.n_add % 2000 == 0:
            self._purge()
        p = t.data_ptr()
        i = bisect.bisect_left(self.starts, p)
        while i < len(self.starts) and self.starts[i] == p:   # a dead tensor used to live here
            del self.starts[i], self.refs[i]
        self.starts.insert(i, p)
        self.refs.insert(i, weakref.ref(t))
        return t

    def view(self, ptr, np_dtype):
        """numpy view (of dtype) starting at ptr up to the end of the owning tensor."""
        if ptr is None or ptr == 0:
            return None
        with self.lock:
            return self._view(ptr, np_dtype)

    def _view(self, ptr, np_dtype):
        i = bisect.bisect_right(self.starts, ptr) - 1
        while i >= 0:
            t = self.refs[i]()
            if t is not None:
                nbytes = t.numel() * t.element_size()
                if self.starts[i] <= ptr < self.starts[i] + nbytes:
                    raw = t.reshape(-1).view(torch.uint8).numpy()
                    raw = raw[ptr - self.starts[i]:]
                    isz = np.dtype(np_dtype).itemsize
                    return raw[:len(raw) // isz * isz].view(np_dtype)
            i -= 1
        raise KeyError("mock_device: pointer %x not owned by a registered tensor" % ptr)


REG = _Registry()


def _npdt(code):
    return np.float64 if code == 0 else np.complex128


def _host(ptr, shape, dtype=np.int64):
    n = int(np.prod(shape))
    buf = (ctypes.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


def _strided(base, off, shape, strides, writeable=False):
    """2-d view base[off + i * strides[0] + j * strides[1]] of a flat array, bounds-checked like the index arrays it replaces
    (element strides, all non-negative in the tables of the kernels)."""
    off, shape, strides = int(off), tuple(int(x) for x in shape), tuple(int(x) for x in strides)
    assert off >= 0 and min(strides) >= 0 and min(shape) >= 1
    last = off + sum((n - 1) * st for n, st in zip(shape, strides))
    if last >= len(base):
        raise IndexError("mock_device: index %d is out of bounds for the arena view of size %d" % (last, len(base)))
    isz = base.itemsize
    return np.lib.stride_tricks.as_strided(base[off:], shape=shape, strides=tuple(st * isz for st in strides), writeable=writeable)


class MockLib:
    def __init__(self):
        self.real = _lib.load()

    def __getattr__(self, name):  # host-only entry points
        if name in ('tpa_svd_theta', 'tpa_svd_theta_store'):
            # device-only composite entry points (csrc/tpa_svd_theta.hip): the emulation runs the Python route of linalg/_svd_warm.py,
            # which is built from emulated primitives (np_conserved._svd_warm_try tests for the attribute)
            raise AttributeError(name)
        return getattr(self.real, name)

    # ---- K1 --------------------------------------------------------------------------------------
    def tpa_gemm_chain(self, code, cfg, tasks_p, links_p, tiles_p, n_tiles, A_p, B_p, C_p, stream):
        dt = _npdt(code)
        tiles = REG.view(tiles_p, np.int32)[:4 * n_tiles].reshape(n_tiles, 4)
        tasks_all = REG.view(tasks_p, np.int64)
        links_all = REG.view(links_p, np.int64)
        A, B, C = REG.view(A_p, dt), REG.view(B_p, dt), REG.view(C_p, dt)
        bm, bn = ctypes.c_int(), ctypes.c_int()
        self.real.tpa_gemm_tile_shape(code, cfg, ctypes.byref(bm), ctypes.byref(bn))
        bm, bn = bm.value, bn.value
        for t, tr, tc, _ in tiles:
            c_off, m, n, ldc, lb, lc, acc, _ = tasks_all[8 * t:8 * t + 8]
            r0, r1 = tr * bm, min(m, (tr + 1) * bm)
            c0, c1 = tc * bn, min(n, (tc + 1) * bn)
            assert r0 < m and c0 < n, "tile outside of task"
            out = np.zeros((r1 - r0, c1 - c0), dtype=dt)
            for l in range(lb, lb + lc):
                a_off, b_off, k, a_rs, a_ks, b_ks, b_ns, flags = links_all[8 * l:8 * l + 8]
                if k <= 0:
                    continue
                Am = _strided(A, a_off + r0 * a_rs, (r1 - r0, k), (a_rs, a_ks))
                Bm = _strided(B, b_off + c0 * b_ns, (k, c1 - c0), (b_ks, b_ns))
                if flags & 1:
                    Am = Am.conj()
                if flags & 2:
                    Bm = Bm.conj()
                out += Am @ Bm
            Cm = _strided(C, c_off + r0 * ldc + c0, out.shape, (ldc, 1), writeable=True)
            if acc:
                Cm += out
            else:
                Cm[...] = out
        return 0

    # ---- K2-K4 -----------------------------------------------------------------------------------
    def tpa_axpy(self, code, n, ar, ai, x_p, y_p, stream):
        dt = _npdt(code)
        al = complex(ar, ai) if code else ar
        x, y = REG.view(x_p, dt)[:n], REG.view(y_p, dt)[:n]
        y += al * x
        return 0

    def tpa_scal(self, code, n, ar, ai, x_p, stream):
        dt = _npdt(code)
        x = REG.view(x_p, dt)[:n]
        x *= (complex(ar, ai) if code else ar)
        return 0

    def tpa_dot(self, code, n, x_p, y_p, do_conj, out_p, scr_p, stream):
        dt = _npdt(code)
        x, y = REG.view(x_p, dt)[:n], REG.view(y_p, dt)[:n]
        v = np.sum((x.conj() if (do_conj and code) else x) * y)
        out = REG.view(out_p, np.float64)
        out[0], out[1] = np.real(v), np.imag(v)
        return 0

    def tpa_nrm2sq(self, code, n, x_p, out_p, scr_p, stream):
        x = REG.view(x_p, _npdt(code))[:n]
        out = REG.view(out_p, np.float64)
        out[0], out[1] = float(np.sum(np.abs(x) ** 2)), 0.
        return 0

    def tpa_lanczos_update(self, code, n, w_p, ar, ai, v1_p, br, bi, v0_p, out_p, scr_p, stream):
        dt = _npdt(code)
        w, v1 = REG.view(w_p, dt)[:n], REG.view(v1_p, dt)[:n]
        w -= (complex(ar, ai) if code else ar) * v1
        if v0_p:
            w -= (complex(br, bi) if code else br) * REG.view(v0_p, dt)[:n]
        out = REG.view(out_p, np.float64)
        out[0], out[1] = float(np.sum(np.abs(w) ** 2)), 0.
        return 0

    # ---- data movement ----------------------------------------------------------------------------
    def tpa_lanczos_step(self, code, n, w_p, v1_p, v0_p, bsq_prev_p, ab_p, scr_p, stream):
        dt = _npdt(code)
        w, v1 = REG.view(w_p, dt)[:n], REG.view(v1_p, dt)[:n]
        ab = REG.view(ab_p, np.float64)
        alpha = float(np.real(np.vdot(w, v1)))
        w -= alpha * v1
        if v0_p:
            w -= np.sqrt(REG.view(bsq_prev_p, np.float64)[0]) * REG.view(v0_p, dt)[:n]
        bsq = float(np.real(np.vdot(w, w)))
        ab[0], ab[1] = alpha, bsq
        if bsq > 0.:
            w *= 1. / np.sqrt(bsq)
        return 0

    _collective = None

    def tpa_lanczos_set_collective(self, cb, user):
        self._collective = cb
        return 0

    def tpa_lanczos_run(self, code, n, ops_p, n_ops, bufs_p, n_bufs, krylov_p, psi0_p, N_max, cutoff, has_shift, E_shift,
                        scal_p, scr_p, cb, user, time_gemms, info_p, stream):
        """Emulation of the C++ loop of tpa_lanczos_run (tenpy_amd/csrc/tpa_vec.hip): same order of device calls, the callback
        one step late."""
        dt = _npdt(code)
        isz = np.dtype(dt).itemsize
        ops = _host(ops_p, (n_ops, 12))
        bufs = _host(bufs_p, (n_bufs,)) if n_bufs else np.zeros(0, np.int64)
        info = _host(info_p, (4,), np.float64)
        psi0 = REG.view(psi0_p, dt)[:n]
        V = lambda k: krylov_p + k * n * isz
        beta0 = float(np.sqrt(np.real(np.vdot(psi0, psi0))))
        info[3] = beta0
        if not beta0 >= cutoff:
            info[0], info[1], info[2] = 0., 1., 0.
            return 0
        REG.view(V(0), dt)[:n] = psi0 / beta0
        hist = {}

        def slot(s, vin, w):
            return vin if s == -1 else (w if s == -2 else int(bufs[s]))
        N, n_mv, stopped = 0, 0, False
        for k in range(N_max):
            vin, w = V(k), V(k + 1)
            for op in ops:
                a, b, c = slot(op[6], vin, w), slot(op[7], vin, w), slot(op[8], vin, w)
                if op[0] == 0:
                    self.tpa_gemm_chain(code, int(op[1]), int(op[2]), int(op[3]), int(op[4]), int(op[5]), a, b, c, stream)
                elif op[0] == 1:
                    self.tpa_lincomb_batch(code, int(op[2]), int(op[5]), int(op[3]), int(op[9]), a, c, stream)
                elif op[0] == 2:
                    self.tpa_copy_batch(code, int(op[2]), int(op[5]), int(op[9]), a, c, stream)
                else:
                    assert op[0] == 3 and self._collective is not None
                    assert self._collective(int(op[1]), None) == 0
            n_mv += 1
            if has_shift:
                self.tpa_axpy(code, n, E_shift, 0., vin, w, stream)
            self.tpa_lanczos_step(code, n, w, vin, V(k - 1) if k > 0 else None, scal_p + 8 * (2 * (k - 1) + 1) if k > 0 else None,
                                  scal_p + 8 * 2 * k, scr_p, stream)
            ab = REG.view(scal_p + 8 * 2 * k, np.float64)
            hist[k] = (float(ab[0]), float(ab[1]))
            if k > 0 and cb(k - 1, hist[k - 1][0], hist[k - 1][1], user):
                N, stopped = k, True
                break
            N = k + 1
        if not stopped:
            cb(N_max - 1, hist[N_max - 1][0], hist[N_max - 1][1], user)
        info[0], info[1], info[2] = N, n_mv, 0.
        return 0

    def tpa_krylov_combine(self, code, n, krylov_p, N, coeff_p, out_p, red_p, scr_p, norm_p, stream):
        dt = _npdt(code)
        coeff = _host(coeff_p, (N,), np.float64)
        V = REG.view(krylov_p, dt)
        out = REG.view(out_p, dt)[:n]
        acc = np.zeros(n, dtype=dt)
        for k in range(N):
            acc += coeff[k] * V[k * n:(k + 1) * n]
        out[:] = acc
        _host(norm_p, (1,), np.float64)[0] = float(np.sqrt(np.real(np.vdot(acc, acc))))
        return 0

    def tpa_copy_batch(self, code, jobs_p, n_jobs, max_elems, src_p, dst_p, stream):
        dt = _npdt(code)
        W = 4 + 3 * MAXD
        jobs = REG.view(jobs_p, np.int64)[:W * n_jobs].reshape(n_jobs, W)
        src, dst = REG.view(src_p, dt), REG.view(dst_p, dt)
        for j in jobs:
            nd = int(j[2])
            shape = tuple(int(x) for x in j[4:4 + nd])
            n = int(np.prod(shape))
            assert n <= max_elems, "max_job_elems too small"
            if n == 0:
                continue
            idx = np.indices(shape).reshape(nd, -1)
            so = j[1] + (idx * j[4 + 2 * MAXD:4 + 2 * MAXD + nd, None]).sum(0)
            do = j[0] + (idx * j[4 + MAXD:4 + MAXD + nd, None]).sum(0)
            v = src[so]
            if code and (j[3] & 1):
                v = v.conj()
            dst[do] = v
        return 0

    def tpa_lincomb_batch(self, code, jobs_p, n_jobs, terms_p, max_elems, src_p, dst_p, stream):
        dt = _npdt(code)
        jobs = REG.view(jobs_p, np.int64)[:8 * n_jobs].reshape(n_jobs, 8)
        n_terms = int(np.max(jobs[:, 4] + jobs[:, 5])) if n_jobs else 0
        terms = REG.view(terms_p, np.int64)[:4 * n_terms].reshape(n_terms, 4)
        alphas = terms[:, 2:4].copy().view(np.float64)
        src, dst = REG.view(src_p, dt), REG.view(dst_p, dt)
        for j in jobs:
            rows, cols = int(j[1]), int(j[2])
            assert rows * cols <= max_elems, "max_job_elems too small"
            r, c = np.indices((rows, cols)).reshape(2, -1)
            acc = np.zeros(rows * cols, dtype=dt)
            for t in range(int(j[4]), int(j[4] + j[5])):
                a = alphas[t, 0] + (1j * alphas[t, 1] if code else 0.)
                acc = acc + a * src[terms[t, 0] + r * terms[t, 1] + c]
            dst[j[0] + r * j[3] + c] = acc
        return 0

    def tpa_tri_lower_batch(self, code, jobs_p, n_jobs, max_elems, g_p, stream):
        dt = _npdt(code)
        jobs = REG.view(jobs_p, np.int64)[:2 * n_jobs].reshape(n_jobs, 2)
        g = REG.view(g_p, dt)
        for g_off, n in jobs:
            blk = g[g_off:g_off + n * n].reshape(n, n)
            d = 0.5 * (np.real(np.diag(blk)) - 1.0)
            blk[:] = np.tril(blk, -1)
            blk[np.arange(n), np.arange(n)] = d
        return 0

    def tpa_scale_axis_batch(self, code, jobs_p, n_jobs, max_elems, x_p, s_p, s_cplx, stream):
        dt = _npdt(code)
        jobs = REG.view(jobs_p, np.int64)[:6 * n_jobs].reshape(n_jobs, 6)
        x = REG.view(x_p, dt)
        s = REG.view(s_p, np.complex128 if s_cplx else np.float64)
        for x_off, pre, ln, post, s_off, _ in jobs:
            blk = x[x_off:x_off + pre * ln * post].reshape(pre, ln, post)
            blk *= s[s_off:s_off + ln][None, :, None]
        return 0

    def tpa_gather_axis_batch(self, code, jobs_p, n_jobs, max_elems, idx_p, src_p, dst_p, stream):
        dt = _npdt(code)
        jobs = REG.view(jobs_p, np.int64)[:8 * n_jobs].reshape(n_jobs, 8)
        idx = REG.view(idx_p, np.int64)
        src, dst = REG.view(src_p, dt), REG.view(dst_p, dt)
        for d_off, s_off, pre, ls, ld, post, i_off, _ in jobs:
            sb = src[s_off:s_off + pre * ls * post].reshape(pre, ls, post)
            dst[d_off:d_off + pre * ld * post] = sb[:, idx[i_off:i_off + ld], :].reshape(-1)
        return 0

    def tpa_axis_sqnorm_batch(self, code, jobs_p, rows_p, n_rows, x_p, out_p, stream):
        rows = REG.view(rows_p, np.int32)[:2 * n_rows].reshape(n_rows, 2)
        jobs_all = REG.view(jobs_p, np.int64)
        x, out = REG.view(x_p, _npdt(code)), REG.view(out_p, np.float64)
        for jb, j in rows:
            if jb < 0:
                continue
            x_off, pre, ln, post, o_off, _ = jobs_all[6 * jb:6 * jb + 6]
            blk = x[x_off:x_off + pre * ln * post].reshape(pre, ln, post)
            out[o_off + j] = float(np.sum(np.abs(blk[:, j, :])**2))
        return 0

    def tpa_convert(self, cf, ct, n, src_p, dst_p, conj, stream):
        src = REG.view(src_p, _npdt(cf))[:n].copy()
        dst = REG.view(dst_p, _npdt(ct))[:n]
        if cf == 1 and ct == 0:
            dst[:] = src.real
        elif cf == 1 and ct == 1 and conj:
            dst[:] = src.conj()
        else:
            dst[:] = src
        return 0

    def tpa_fill_zero(self, dst_p, nbytes, stream):
        REG.view(dst_p, np.uint8)[:nbytes] = 0
        return 0

    # ---- factorizations (LAPACK stands in for the Jacobi / Householder kernels) --------------------
    def tpa_svd_worksize(self, code, jobs_p, n):
        return 256

    def tpa_svd_batch(self, code, jobs_p, n_jobs, a_p, u_p, s_p, vh_p, work_p, wb, max_sweeps, tol, sweeps_p, stream):
        dt = _npdt(code)
        jobs = _host(jobs_p, (n_jobs, 8))
        A, U, VH = REG.view(a_p, dt), REG.view(u_p, dt), REG.view(vh_p, dt)
        S = REG.view(s_p, np.float64)
        for a_off, m, n, u_off, s_off, v_off, _, _ in jobs:
            k = min(m, n)
            blk = A[a_off:a_off + m * n].reshape(m, n)
            if not np.all(np.isfinite(blk)):
                return _lib.E_NAN
            u, s, vh = scipy.linalg.svd(blk, full_matrices=False, lapack_driver='gesvd')
            U[u_off:u_off + m * k] = u.reshape(-1)
            S[s_off:s_off + k] = s
            VH[v_off:v_off + k * n] = vh.reshape(-1)
        return 0

    def tpa_qr_batch(self, code, jobs_p, n_jobs, a_p, q_p, r_p, stream):
        dt = _npdt(code)
        jobs = _host(jobs_p, (n_jobs, 8))
        Q, R = REG.view(q_p, dt), REG.view(r_p, dt)
        isz = np.dtype(dt).itemsize
        for a_off, m, n, q_off, r_off, _, _, _ in jobs:
            k = min(m, n)
            A = REG.view(a_p + int(a_off) * isz, dt)      # (np_conserved.qr_batched: the blocks of several arenas, offsets from one base address)
            q, r = np.linalg.qr(A[:m * n].reshape(m, n), mode='reduced')
            Q[q_off:q_off + m * k] = q.reshape(-1)
            R[r_off:r_off + k * n] = r.reshape(-1)
        return 0

    def tpa_eigh_worksize(self, code, jobs_p, n):
        return 256

    def tpa_eigh_set_direct(self, on):
        return 0

    def tpa_eigh_from_svd(self, code, jobs_p, n_jobs, u_p, s_p, vh_p, lam_p, err_p, stream):
        dt = _npdt(code)
        jobs = _host(jobs_p, (n_jobs, 8))
        U, VH = REG.view(u_p, dt), REG.view(vh_p, dt)
        S, lam, err = REG.view(s_p, np.float64), REG.view(lam_p, np.float64), REG.view(err_p, np.float64)
        for b, (u_off, n, s_off, vh_off, lam_off, _, _, _) in enumerate(jobs):
            u = U[u_off:u_off + n * n].reshape(n, n)
            v = VH[vh_off:vh_off + n * n].reshape(n, n).conj().T
            sg = S[s_off:s_off + n].copy()
            d = np.where(np.real(np.sum(u.conj() * v, axis=0)) < 0, -1., 1.)
            lam[lam_off:lam_off + n] = d * sg
            err[b] = np.max(sg * np.linalg.norm(v - u * d[None, :], axis=0)) if n else 0.
        return 0

    def tpa_eigh_batch(self, code, jobs_p, n_jobs, a_p, w_p, v_p, work_p, wb, max_sweeps, tol, sweeps_p, stream):
        dt = _npdt(code)
        jobs = _host(jobs_p, (n_jobs, 8))
        A, V = REG.view(a_p, dt), REG.view(v_p, dt)
        W = REG.view(w_p, np.float64)
        for a_off, n, w_off, v_off, _, _, _, _ in jobs:
            w, v = np.linalg.eigh(A[a_off:a_off + n * n].reshape(n, n), 'L')
            W[w_off:w_off + n] = w
            V[v_off:v_off + n * n] = v.reshape(-1)
        return 0


class _FakeCuda:
    @staticmethod
    def current_device():
        return 0


def install(monkeypatch):
    """Patch ``tenpy_amd.linalg._device`` so that arenas are CPU tensors and kernels are emulated."""
    mock = MockLib()

    def empty(n, dtype):
        return REG.add(torch.empty(int(n), dtype=dev.tdtype(dtype)))

    def zeros(n, dtype):
        return REG.add(torch.zeros(int(n), dtype=dev.tdtype(dtype)))

    def to_device(arr):
        return REG.add(torch.from_numpy(np.array(arr, copy=True, order='C')))

    def reduction_buffers():
        import threading
        key = ('mock', threading.get_ident())      # per thread, like _device.reduction_buffers (tests/test_threads.py)
        if key not in dev._scratch:
            dev._scratch[key] = (REG.add(torch.zeros(4, dtype=torch.float64)),
                                 REG.add(torch.zeros(4096, dtype=torch.float64)))
        return dev._scratch[key]

    class _T:
        """torch facade: allocation helpers register their tensors."""
        uint8 = torch.uint8
        float64 = torch.float64
        complex128 = torch.complex128

        @staticmethod
        def empty(n, dtype=None, device=None):
            return REG.add(torch.empty(int(n), dtype=dtype))

        class cuda:
            """Just enough of torch.cuda for the event timer and the phase timers (host clock)."""
            class Event:
                def __init__(self, enable_timing=True):
                    self.t = None

                def record(self):
                    import time
                    self.t = time.perf_counter()

                def synchronize(self):
                    pass

                def elapsed_time(self, other):
                    return 1e3 * (other.t - self.t)

            @staticmethod
            def synchronize(*a, **k):
                pass

            @staticmethod
            def current_device():
                return 0

    class ScalarPipe:
        """Emulation of dev.ScalarPipe: the 'device' buffer is a registered CPU tensor, reads are immediate."""

        def __init__(self, n_steps):
            self.dev = REG.add(torch.zeros((n_steps, 2), dtype=torch.float64))

        def ptr(self, step, which=0):
            return self.dev.data_ptr() + 8 * (2 * step + which)

        def post(self, step):
            pass

        def get(self, step):
            return float(self.dev[step, 0]), float(self.dev[step, 1])

    monkeypatch.setattr(dev, "ScalarPipe", ScalarPipe)
    monkeypatch.setattr(dev, "empty", empty)
    monkeypatch.setattr(dev, "zeros", zeros)
    monkeypatch.setattr(dev, "to_device", to_device)
    monkeypatch.setattr(dev, "clone", lambda t: REG.add(t.clone()))
    monkeypatch.setattr(dev, "lib", lambda: mock)
    monkeypatch.setattr(dev, "stream", lambda: 0)
    monkeypatch.setattr(dev, "reduction_buffers", reduction_buffers)
    monkeypatch.setattr(dev, "torch", lambda: _T)
    return mock
