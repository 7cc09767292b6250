// K2-K4: vector kernels over packed block arenas (HBM-bound), gfx950.
//
// An Array's blocks are packed back to back in one arena, so when two Arrays share the same block
// structure (always true inside a Lanczos run) the block-wise BLAS-1 loops of the reference
// (_inner_worker _npc_helper.pyx:1791-1875, _blas_inpl_add :316, _blas_inpl_scale :339,
// Array.norm np_conserved.py:2241-2255) collapse to ONE flat pass over the arena.
// Reductions are deterministic: pass 1 writes one partial per workgroup, pass 2 (one workgroup)
// sums them in a fixed order.
#include "tpa_common.h"
#include <vector>

namespace {

constexpr int NT = 256;
constexpr int MAXBLK = 1024;  // partials per reduction (<= TPA_RED_SCRATCH / 2)

__host__ inline int grid_for(int64_t n, int per_thread) {
    int64_t g = (n + (int64_t)NT * per_thread - 1) / ((int64_t)NT * per_thread);
    if (g < 1) g = 1;
    if (g > MAXBLK) g = MAXBLK;
    return (int)g;
}

// ---- y += alpha x ---------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void axpy_f64(int64_t n, double a, const double *__restrict__ x,
                                               double *__restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT)
        y[i] = fma(a, x[i], y[i]);
}
__global__ __launch_bounds__(NT) void axpy_c128(int64_t n, double ar, double ai,
                                                const double2 *__restrict__ x,
                                                double2 *__restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        double2 xv = x[i], yv = y[i];
        yv.x += ar * xv.x - ai * xv.y;
        yv.y += ar * xv.y + ai * xv.x;
        y[i] = yv;
    }
}
__global__ __launch_bounds__(NT) void scal_f64(int64_t n, double a, double *__restrict__ x) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT)
        x[i] *= a;
}
__global__ __launch_bounds__(NT) void scal_c128(int64_t n, double ar, double ai,
                                                double2 *__restrict__ x) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        double2 v = x[i];
        x[i] = double2{ar * v.x - ai * v.y, ar * v.y + ai * v.x};
    }
}

// ---- reductions -------------------------------------------------------------------------------
// mode 0: dot(x,y) real; 1: sum conj(x) y (complex); 2: sum x y (complex, no conj); 3: sum |x|^2
template <int MODE>
__global__ __launch_bounds__(NT) void reduce_pass1(int64_t n, const double *__restrict__ x,
                                                   const double *__restrict__ y,
                                                   double *__restrict__ partial) {
    __shared__ double red[NT / 64];
    double sr = 0, si = 0;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        if (MODE == 0) {
            sr = fma(x[i], y[i], sr);
        } else if (MODE == 3) {
            sr = fma(x[i], x[i], sr);
        } else {
            const double2 a = reinterpret_cast<const double2 *>(x)[i];
            const double2 b = reinterpret_cast<const double2 *>(y)[i];
            if (MODE == 1) {  // conj(a) * b
                sr += a.x * b.x + a.y * b.y;
                si += a.x * b.y - a.y * b.x;
            } else {
                sr += a.x * b.x - a.y * b.y;
                si += a.x * b.y + a.y * b.x;
            }
        }
    }
    sr = block_sum<NT>(sr, red);
    if (MODE == 1 || MODE == 2) si = block_sum<NT>(si, red);
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = sr;
        partial[2 * blockIdx.x + 1] = si;
    }
}

__global__ __launch_bounds__(NT) void reduce_pass2(int nblk, const double *__restrict__ partial,
                                                   double *__restrict__ out) {
    __shared__ double red[NT / 64];
    double sr = 0, si = 0;
    for (int i = threadIdx.x; i < nblk; i += NT) {
        sr += partial[2 * i];
        si += partial[2 * i + 1];
    }
    sr = block_sum<NT>(sr, red);
    si = block_sum<NT>(si, red);
    if (threadIdx.x == 0) {
        out[0] = sr;
        out[1] = si;
    }
}

// ---- fused Lanczos three-term recurrence: w -= a v1 ; w -= b v0 ; partial |w|^2 ----------------
template <bool CPLX, bool HAVE_V0>
__global__ __launch_bounds__(NT) void lanczos_update_kernel(int64_t n, double *__restrict__ w,
                                                            double ar, double ai,
                                                            const double *__restrict__ v1, double br,
                                                            double bi, const double *__restrict__ v0,
                                                            double *__restrict__ partial) {
    __shared__ double red[NT / 64];
    double s = 0;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        if (!CPLX) {
            double t = w[i];
            t = fma(-ar, v1[i], t);
            if (HAVE_V0) t = fma(-br, v0[i], t);
            w[i] = t;
            s = fma(t, t, s);
        } else {
            double2 t = reinterpret_cast<double2 *>(w)[i];
            const double2 p = reinterpret_cast<const double2 *>(v1)[i];
            t.x -= ar * p.x - ai * p.y;
            t.y -= ar * p.y + ai * p.x;
            if (HAVE_V0) {
                const double2 q = reinterpret_cast<const double2 *>(v0)[i];
                t.x -= br * q.x - bi * q.y;
                t.y -= br * q.y + bi * q.x;
            }
            reinterpret_cast<double2 *>(w)[i] = t;
            s += t.x * t.x + t.y * t.y;
        }
    }
    s = block_sum<NT>(s, red);
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = s;
        partial[2 * blockIdx.x + 1] = 0;
    }
}

// ---- one Lanczos step with device-resident scalars (no host round trip between the kernels) ----------------------------
// alpha comes from ab[0] (written by the dot reduction just before), beta_prev = sqrt(bsq_prev[0]) from the previous step.
template <bool CPLX, bool HAVE_V0>
__global__ __launch_bounds__(NT) void lanczos_update_dev_kernel(int64_t n, double *__restrict__ w,
                                                                const double *__restrict__ ab,
                                                                const double *__restrict__ v1,
                                                                const double *__restrict__ bsq_prev,
                                                                const double *__restrict__ v0,
                                                                double *__restrict__ partial) {
    __shared__ double red[NT / 64];
    const double ar = ab[0];
    const double br = HAVE_V0 ? sqrt(bsq_prev[0]) : 0.0;
    double s = 0;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        if (!CPLX) {
            double t = w[i];
            t = fma(-ar, v1[i], t);
            if (HAVE_V0) t = fma(-br, v0[i], t);
            w[i] = t;
            s = fma(t, t, s);
        } else {
            double2 t = reinterpret_cast<double2 *>(w)[i];
            const double2 p = reinterpret_cast<const double2 *>(v1)[i];
            t.x -= ar * p.x;
            t.y -= ar * p.y;
            if (HAVE_V0) {
                const double2 q = reinterpret_cast<const double2 *>(v0)[i];
                t.x -= br * q.x;
                t.y -= br * q.y;
            }
            reinterpret_cast<double2 *>(w)[i] = t;
            s += t.x * t.x + t.y * t.y;
        }
    }
    s = block_sum<NT>(s, red);
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = s;
        partial[2 * blockIdx.x + 1] = 0;
    }
}

// second pass of the norm: out[1] = sum of the partials (out[0] keeps alpha)
__global__ __launch_bounds__(NT) void reduce_pass2_to(int nblk, const double *__restrict__ partial, double *__restrict__ out) {
    __shared__ double red[NT / 64];
    double sr = 0;
    for (int i = threadIdx.x; i < nblk; i += NT) sr += partial[2 * i];
    sr = block_sum<NT>(sr, red);
    if (threadIdx.x == 0) out[0] = sr;
}

// x *= 1 / sqrt(bsq[0])  (x as flat doubles; nothing happens for bsq <= 0)
__global__ __launch_bounds__(NT) void scal_rsqrt_dev_kernel(int64_t nd, double *__restrict__ x, const double *__restrict__ bsq) {
    const double b = bsq[0];
    if (!(b > 0.0)) return;
    const double f = 1.0 / sqrt(b);
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < nd; i += (int64_t)gridDim.x * NT) x[i] *= f;
}

}  // namespace

extern "C" int tpa_axpy(int dtype, int64_t n, double ar, double ai, const void *x, void *y,
                        void *stream) {
    TPA_ARG_CHECK(dtype == TPA_F64 || dtype == TPA_C128);
    if (n <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int g = grid_for(n, 4) * 2;
    if (dtype == TPA_F64)
        axpy_f64<<<g, NT, 0, st>>>(n, ar, (const double *)x, (double *)y);
    else
        axpy_c128<<<g, NT, 0, st>>>(n, ar, ai, (const double2 *)x, (double2 *)y);
    TPA_LAUNCH_CHECK();
    return 0;
}

extern "C" int tpa_scal(int dtype, int64_t n, double ar, double ai, void *x, void *stream) {
    TPA_ARG_CHECK(dtype == TPA_F64 || dtype == TPA_C128);
    if (n <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int g = grid_for(n, 4) * 2;
    if (dtype == TPA_F64)
        scal_f64<<<g, NT, 0, st>>>(n, ar, (double *)x);
    else
        scal_c128<<<g, NT, 0, st>>>(n, ar, ai, (double2 *)x);
    TPA_LAUNCH_CHECK();
    return 0;
}

extern "C" int tpa_dot(int dtype, int64_t n, const void *x, const void *y, int do_conj,
                       double *out, double *scratch, void *stream) {
    TPA_ARG_CHECK(dtype == TPA_F64 || dtype == TPA_C128);
    hipStream_t st = (hipStream_t)stream;
    const int g = (n > 0) ? grid_for(n, 8) : 1;
    if (dtype == TPA_F64)
        reduce_pass1<0><<<g, NT, 0, st>>>(n, (const double *)x, (const double *)y, scratch);
    else if (do_conj)
        reduce_pass1<1><<<g, NT, 0, st>>>(n, (const double *)x, (const double *)y, scratch);
    else
        reduce_pass1<2><<<g, NT, 0, st>>>(n, (const double *)x, (const double *)y, scratch);
    TPA_LAUNCH_CHECK();
    reduce_pass2<<<1, NT, 0, st>>>(g, scratch, out);
    TPA_LAUNCH_CHECK();
    return 0;
}

extern "C" int tpa_nrm2sq(int dtype, int64_t n, const void *x, double *out, double *scratch,
                          void *stream) {
    TPA_ARG_CHECK(dtype == TPA_F64 || dtype == TPA_C128);
    hipStream_t st = (hipStream_t)stream;
    const int64_t nd = (dtype == TPA_C128) ? 2 * n : n;  // |z|^2 = re^2 + im^2: flat real pass
    const int g = (nd > 0) ? grid_for(nd, 8) : 1;
    reduce_pass1<3><<<g, NT, 0, st>>>(nd, (const double *)x, (const double *)x, scratch);
    TPA_LAUNCH_CHECK();
    reduce_pass2<<<1, NT, 0, st>>>(g, scratch, out);
    TPA_LAUNCH_CHECK();
    return 0;
}

extern "C" int tpa_lanczos_update(int dtype, int64_t n, void *w, double ar, double ai,
                                  const void *v1, double br, double bi, const void *v0, double *out,
                                  double *scratch, void *stream) {
    TPA_ARG_CHECK(dtype == TPA_F64 || dtype == TPA_C128);
    TPA_ARG_CHECK(v1 != nullptr);
    hipStream_t st = (hipStream_t)stream;
    const int g = (n > 0) ? grid_for(n, 8) : 1;
    if (dtype == TPA_F64) {
        if (v0)
            lanczos_update_kernel<false, true><<<g, NT, 0, st>>>(n, (double *)w, ar, ai, (const double *)v1, br, bi, (const double *)v0, scratch);
        else
            lanczos_update_kernel<false, false><<<g, NT, 0, st>>>(n, (double *)w, ar, ai, (const double *)v1, br, bi, nullptr, scratch);
    } else {
        if (v0)
            lanczos_update_kernel<true, true><<<g, NT, 0, st>>>(n, (double *)w, ar, ai, (const double *)v1, br, bi, (const double *)v0, scratch);
        else
            lanczos_update_kernel<true, false><<<g, NT, 0, st>>>(n, (double *)w, ar, ai, (const double *)v1, br, bi, nullptr, scratch);
    }
    TPA_LAUNCH_CHECK();
    reduce_pass2<<<1, NT, 0, st>>>(g, scratch, out);
    TPA_LAUNCH_CHECK();
    return 0;
}

extern "C" int tpa_fill_zero(void *dst, int64_t n_bytes, void *stream) {
    if (n_bytes <= 0) return 0;
    TPA_HIP_CHECK(hipMemsetAsync(dst, 0, (size_t)n_bytes, (hipStream_t)stream));
    return 0;
}

// One Lanczos step of krylov_based.py:655-672 with the scalars kept on the device, so that the host can enqueue the next
// matvec without waiting:   alpha = Re <w|v1> ;  w -= alpha v1 ;  w -= beta_prev v0 ;  bsq = |w|^2 ;  w /= sqrt(bsq).
// ab_out[0] = alpha, ab_out[1] = bsq (= beta^2 of this step; the NEXT step passes &ab_out[1] as bsq_prev).
extern "C" int tpa_lanczos_step(int dtype, int64_t n, void *w, const void *v1, const void *v0,
                                const double *bsq_prev, double *ab_out, double *scratch, void *stream) {
    TPA_ARG_CHECK(dtype == TPA_F64 || dtype == TPA_C128);
    TPA_ARG_CHECK(v1 != nullptr && ab_out != nullptr && (v0 == nullptr || bsq_prev != nullptr));
    hipStream_t st = (hipStream_t)stream;
    const bool cplx = (dtype == TPA_C128);
    const int g = (n > 0) ? grid_for(n, 8) : 1;
    if (!cplx)
        reduce_pass1<0><<<g, NT, 0, st>>>(n, (const double *)w, (const double *)v1, scratch);
    else
        reduce_pass1<1><<<g, NT, 0, st>>>(n, (const double *)w, (const double *)v1, scratch);
    reduce_pass2<<<1, NT, 0, st>>>(g, scratch, ab_out);          // ab_out[0] = Re <w|v1>, ab_out[1] = Im (overwritten below)
    double *part2 = scratch + 2 * MAXBLK;
    if (!cplx) {
        if (v0)
            lanczos_update_dev_kernel<false, true><<<g, NT, 0, st>>>(n, (double *)w, ab_out, (const double *)v1, bsq_prev, (const double *)v0, part2);
        else
            lanczos_update_dev_kernel<false, false><<<g, NT, 0, st>>>(n, (double *)w, ab_out, (const double *)v1, nullptr, nullptr, part2);
    } else {
        if (v0)
            lanczos_update_dev_kernel<true, true><<<g, NT, 0, st>>>(n, (double *)w, ab_out, (const double *)v1, bsq_prev, (const double *)v0, part2);
        else
            lanczos_update_dev_kernel<true, false><<<g, NT, 0, st>>>(n, (double *)w, ab_out, (const double *)v1, nullptr, nullptr, part2);
    }
    reduce_pass2_to<<<1, NT, 0, st>>>(g, part2, ab_out + 1);
    const int64_t nd = cplx ? 2 * n : n;
    scal_rsqrt_dev_kernel<<<grid_for(nd, 8), NT, 0, st>>>(nd, (double *)w, ab_out + 1);
    TPA_LAUNCH_CHECK();
    return 0;
}

// ================================================================================================================
// LanczosGroundState.run as ONE host call (reference krylov_based.py:645-700 `_build_krylov`, :223-240 `_calc_result_full`).
//
// Why: at chi <= 512 a bond update is bounded by host time, not by the kernels -- every Krylov step costs ~0.5 ms of
// interpreter work (Array objects, argument checks, ctypes marshalling for the 2-3 launches of a matvec and the 5 of
// the recurrence) against ~0.1 ms of kernels.  Here the host side of a step is a C++ loop: replay the matvec "program"
// (the cached plans of TwoSiteH: grouped GEMM / block linear combination / grouped GEMM) on raw arenas, run the fused
// recurrence with alpha, beta kept on the device, and hand (alpha, beta^2) of the PREVIOUS step to a callback that does
// the reference's tridiagonal eigen-solve and stopping test (numpy.linalg.eigh, so that iteration counts and signs are
// those of the reference) while the device is already busy with the next matvec.
// ================================================================================================================
namespace {

constexpr int LZ_MAX_COMBINE = 64;
struct LzCoeff {
    double c[LZ_MAX_COMBINE];
};

// out = sum_k c_k V_k (V_k = krylov + k n), partial |out|^2 per workgroup: one pass over N + 1 vectors instead of N axpys
template <bool CPLX>
__global__ __launch_bounds__(NT) void krylov_combine_kernel(int64_t nd, int N, LzCoeff C, const double *__restrict__ V, int64_t stride,
                                                            double *__restrict__ out, double *__restrict__ partial) {
    __shared__ double red[NT / 64];
    double s = 0;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < nd; i += (int64_t)gridDim.x * NT) {
        double t = 0;
        for (int k = 0; k < N; ++k) t = fma(C.c[k], V[k * stride + i], t);
        out[i] = t;
        s = fma(t, t, s);
    }
    s = block_sum<NT>(s, red);
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = s;
        partial[2 * blockIdx.x + 1] = 0.;
    }
}

__global__ void lz_post_scalars_kernel(const double *__restrict__ src, double *__restrict__ dst_host) {
    dst_host[0] = src[0];
    dst_host[1] = src[1];
    __threadfence_system();
}

__global__ __launch_bounds__(NT) void lz_copy_scaled_kernel(int64_t nd, const double *__restrict__ x, const double *__restrict__ nrm2,
                                                            double *__restrict__ y) {
    const double f = 1. / sqrt(nrm2[0]);
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < nd; i += (int64_t)gridDim.x * NT) y[i] = f * x[i];
}

struct LzHost {          // per-thread persistent host resources
    double *pinned = nullptr;        // mapped host memory: 2 doubles per step
    int capacity = 0;
    std::vector<hipEvent_t> ev;      // one per step (no timing)
    std::vector<hipEvent_t> tev;     // timing events for the GEMM launches (only when asked for)
};
thread_local LzHost lz_host;

int lz_reserve(int steps) {
    LzHost &H = lz_host;
    if (H.capacity < steps + 2) {
        if (H.pinned) TPA_HIP_CHECK(hipHostFree(H.pinned));
        H.capacity = 2 * (steps + 2);
        TPA_HIP_CHECK(hipHostMalloc((void **)&H.pinned, sizeof(double) * 2 * H.capacity, hipHostMallocDefault));
    }
    while ((int)H.ev.size() < steps + 2) {
        hipEvent_t e;
        TPA_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        H.ev.push_back(e);
    }
    return 0;
}

inline void *lz_slot(int64_t s, void *const *bufs, int n_bufs, const void *in, void *out) {
    if (s == -1) return const_cast<void *>(in);
    if (s == -2) return out;
    return (s >= 0 && s < n_bufs) ? bufs[s] : nullptr;
}

}  // namespace

// collective hook of the launch programs (op kind 3): set by the caller around a run, per host thread
static thread_local tpa_collective_callback lz_collective = nullptr;
static thread_local void *lz_collective_user = nullptr;
extern "C" int tpa_lanczos_set_collective(tpa_collective_callback cb, void *user) {
    lz_collective = cb;
    lz_collective_user = user;
    return 0;
}

extern "C" int tpa_lanczos_run(int dtype, int64_t n, const int64_t *ops, int n_ops, void *const *bufs, int n_bufs,
                               void *krylov_dev, const void *psi0_dev, int N_max, double cutoff, int has_shift, double E_shift,
                               double *scalars_dev, double *scratch_dev, tpa_lanczos_callback cb, void *user,
                               int time_gemms, double *info, void *stream) {
    TPA_ARG_CHECK(dtype == TPA_F64 || dtype == TPA_C128);
    TPA_ARG_CHECK(n > 0 && N_max >= 1 && n_ops >= 1 && ops != nullptr && krylov_dev != nullptr && psi0_dev != nullptr);
    TPA_ARG_CHECK(scalars_dev != nullptr && scratch_dev != nullptr && cb != nullptr && info != nullptr);
    hipStream_t st = (hipStream_t)stream;
    const bool cplx = (dtype == TPA_C128);
    const int64_t nd = cplx ? 2 * n : n;
    double *V = (double *)krylov_dev;
    if (int rc = lz_reserve(N_max)) return rc;
    LzHost &H = lz_host;
    // beta_0 = |psi0| ; V_0 = psi0 / beta_0     (reference :650-653)
    double *sc0 = scalars_dev + 2 * (N_max + 1);        // |psi0|^2 lives behind the per-step scalars
    if (int rc = tpa_nrm2sq(dtype, n, psi0_dev, sc0, scratch_dev, stream)) return rc;
    lz_copy_scaled_kernel<<<grid_for(nd, 8), NT, 0, st>>>(nd, (const double *)psi0_dev, sc0, V);
    lz_post_scalars_kernel<<<1, 1, 0, st>>>(sc0, H.pinned + 2 * (N_max + 1));
    TPA_HIP_CHECK(hipEventRecord(H.ev[N_max + 1], st));
    size_t n_tev = 0;
    int n_matvec = 0, N = 0;
    bool stopped = false;
    for (int k = 0; k < N_max; ++k) {
        const double *vin = V + (int64_t)k * nd;
        double *w = V + (int64_t)(k + 1) * nd;
        for (int o = 0; o < n_ops; ++o) {
            const int64_t *op = ops + 12 * o;
            void *a = lz_slot(op[6], bufs, n_bufs, vin, w), *b = lz_slot(op[7], bufs, n_bufs, vin, w), *c = lz_slot(op[8], bufs, n_bufs, vin, w);
            if (op[0] == 0) {
                TPA_ARG_CHECK(a != nullptr && b != nullptr && c != nullptr);
                if (time_gemms) {
                    while (H.tev.size() < n_tev + 2) {
                        hipEvent_t e;
                        TPA_HIP_CHECK(hipEventCreate(&e));
                        H.tev.push_back(e);
                    }
                    TPA_HIP_CHECK(hipEventRecord(H.tev[n_tev], st));
                }
                if (int rc = tpa_gemm_chain(dtype, (int)op[1], (const int64_t *)op[2], (const int64_t *)op[3], (const int32_t *)op[4], (int)op[5],
                                            a, b, c, stream))
                    return rc;
                if (time_gemms) {
                    TPA_HIP_CHECK(hipEventRecord(H.tev[n_tev + 1], st));
                    n_tev += 2;
                }
            } else if (op[0] == 1) {
                TPA_ARG_CHECK(a != nullptr && c != nullptr);
                if (int rc = tpa_lincomb_batch(dtype, (const int64_t *)op[2], (int)op[5], (const int64_t *)op[3], op[9], a, c, stream)) return rc;
                // cfg = 1: the reduction of the split-K partial blocks of the GEMM just launched -- part of that GEMM's time
                if (time_gemms && op[1] == 1 && n_tev >= 2) TPA_HIP_CHECK(hipEventRecord(H.tev[n_tev - 1], st));
            } else if (op[0] == 2) {        // batched strided copy (pack / unpack of the row panels of a sharded matvec)
                TPA_ARG_CHECK(a != nullptr && c != nullptr);
                if (int rc = tpa_copy_batch(dtype, (const int64_t *)op[2], (int)op[5], op[9], a, c, stream)) return rc;
            } else if (op[0] == 3) {        // collective of the caller (all-gather of the row panels): enqueued by the host callback
                TPA_ARG_CHECK(lz_collective != nullptr);
                if (int rc = lz_collective((int)op[1], lz_collective_user)) {
                    snprintf(tpa_errbuf, sizeof(tpa_errbuf), "tpa_lanczos_run: the collective callback of op %d failed (%d)", o, rc);
                    return TPA_E_BADARG;
                }
            } else {
                TPA_ARG_CHECK(false && "unknown op kind");
            }
        }
        ++n_matvec;
        if (has_shift)
            if (int rc = tpa_axpy(dtype, n, E_shift, 0., vin, w, stream)) return rc;
        if (int rc = tpa_lanczos_step(dtype, n, w, vin, k > 0 ? (const void *)(V + (int64_t)(k - 1) * nd) : nullptr,
                                      k > 0 ? scalars_dev + 2 * (k - 1) + 1 : nullptr, scalars_dev + 2 * k, scratch_dev, stream))
            return rc;
        lz_post_scalars_kernel<<<1, 1, 0, st>>>(scalars_dev + 2 * k, H.pinned + 2 * k);
        TPA_HIP_CHECK(hipEventRecord(H.ev[k], st));
        if (k == 0) {       // |psi0| (reference :650: "norm of psi0 too small" is the caller's error; step 0 is in flight meanwhile)
            TPA_HIP_CHECK(hipEventSynchronize(H.ev[N_max + 1]));
            info[3] = sqrt(H.pinned[2 * (N_max + 1)]);
            if (!(info[3] >= cutoff)) {
                info[0] = 0.;
                info[1] = 1.;
                info[2] = 0.;
                TPA_HIP_CHECK(hipStreamSynchronize(st));
                return 0;
            }
        }
        if (k > 0) {
            TPA_HIP_CHECK(hipEventSynchronize(H.ev[k - 1]));
            if (cb(k - 1, H.pinned[2 * (k - 1)], H.pinned[2 * (k - 1) + 1], user)) {
                N = k;      // step k in flight is not part of the result (same Krylov space as the step-by-step loop)
                stopped = true;
                break;
            }
        }
        N = k + 1;
    }
    if (!stopped) {
        TPA_HIP_CHECK(hipEventSynchronize(H.ev[N_max - 1]));
        cb(N_max - 1, H.pinned[2 * (N_max - 1)], H.pinned[2 * (N_max - 1) + 1], user);
    }
    double ms = 0.;
    if (time_gemms && n_tev) {
        TPA_HIP_CHECK(hipEventSynchronize(H.tev[n_tev - 1]));
        for (size_t i = 0; i < n_tev; i += 2) {
            float t = 0.f;
            TPA_HIP_CHECK(hipEventElapsedTime(&t, H.tev[i], H.tev[i + 1]));
            ms += t;
        }
    }
    info[0] = (double)N;
    info[1] = (double)n_matvec;
    info[2] = ms;
    return 0;
}

// psi = sum_k coeff_k V_k, then |psi| (returned in out_host[0], blocking) -- the first half of `_calc_result_full`
// (krylov_based.py:223-236); the caller decides about the ill-conditioned / degenerate cases and scales.
extern "C" int tpa_krylov_combine(int dtype, int64_t n, const void *krylov_dev, int N, const double *coeff, void *out_dev,
                                  double *red_out_dev, double *scratch_dev, double *norm_host, void *stream) {
    TPA_ARG_CHECK(dtype == TPA_F64 || dtype == TPA_C128);
    TPA_ARG_CHECK(n > 0 && N >= 1 && N <= LZ_MAX_COMBINE && coeff != nullptr && out_dev != nullptr && norm_host != nullptr);
    hipStream_t st = (hipStream_t)stream;
    const bool cplx = (dtype == TPA_C128);
    const int64_t nd = cplx ? 2 * n : n;
    LzCoeff C;
    for (int k = 0; k < LZ_MAX_COMBINE; ++k) C.c[k] = k < N ? coeff[k] : 0.;
    const int g = grid_for(nd, 4);
    if (cplx)
        krylov_combine_kernel<true><<<g, NT, 0, st>>>(nd, N, C, (const double *)krylov_dev, nd, (double *)out_dev, scratch_dev);
    else
        krylov_combine_kernel<false><<<g, NT, 0, st>>>(nd, N, C, (const double *)krylov_dev, nd, (double *)out_dev, scratch_dev);
    reduce_pass2<<<1, NT, 0, st>>>(g, scratch_dev, red_out_dev);
    TPA_LAUNCH_CHECK();
    if (int rc = lz_reserve(1)) return rc;
    lz_post_scalars_kernel<<<1, 1, 0, st>>>(red_out_dev, lz_host.pinned);
    TPA_HIP_CHECK(hipEventRecord(lz_host.ev[0], st));
    TPA_HIP_CHECK(hipEventSynchronize(lz_host.ev[0]));
    norm_host[0] = sqrt(lz_host.pinned[0]);
    return 0;
}
