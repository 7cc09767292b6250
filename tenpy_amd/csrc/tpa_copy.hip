// K8/K9/K10: data-movement kernels (HBM-bound), gfx950.
//
//  * tpa_copy_batch      : batched N-d strided sub-block copy.  One job = one (old block -> slice of
//                          new block) memcpy of the reference's combine/split workers
//                          (_npc_helper.pyx:1112-1123, :1235-1240 via _sliced_strided_copy :368) or one
//                          per-block transpose of itranspose (:853).  The host builds the copy plan
//                          from the integer bookkeeping; the device only moves bytes.
//  * tpa_scale_axis_batch: iscale_axis (np_conserved.py:2132-2140).
//  * tpa_gather_axis_batch: iproject's np.compress along one axis (np_conserved.py:1982).
#include "tpa_common.h"

namespace {
constexpr int NT = 256;
constexpr int MAXD = TPA_COPY_MAXDIM;

struct CopyJob {  // int64[4 + 3*MAXD]
    int64_t dst_off, src_off, ndim, flags;
    int64_t shape[MAXD], dstr[MAXD], sstr[MAXD];
};

template <bool CPLX>
__global__ __launch_bounds__(NT) void copy_batch_kernel(const CopyJob *__restrict__ jobs,
                                                        const double *__restrict__ src,
                                                        double *__restrict__ dst) {
    const CopyJob &J = jobs[blockIdx.y];
    const int nd = (int)J.ndim;
    int64_t total = 1;
    for (int d = 0; d < nd; ++d) total *= J.shape[d];
    const bool conj = CPLX && (J.flags & 1);
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < total; e += (int64_t)gridDim.x * NT) {
        int64_t rem = e, so = J.src_off, dof = J.dst_off;
        for (int d = nd - 1; d >= 0; --d) {
            const int64_t s = J.shape[d];
            const int64_t q = rem / s;
            const int64_t i = rem - q * s;
            rem = q;
            so += i * J.sstr[d];
            dof += i * J.dstr[d];
        }
        if (CPLX) {
            double2 v = reinterpret_cast<const double2 *>(src)[so];
            if (conj) v.y = -v.y;
            reinterpret_cast<double2 *>(dst)[dof] = v;
        } else {
            dst[dof] = src[so];
        }
    }
}


// dst slab = sum_t alpha_t * src_t slab  (row-major slabs with their own row strides; coalesced along the columns)
struct LinJob {   // int64[8]
    int64_t dst_off, rows, cols, dst_ld, term_begin, term_count, pad0, pad1;
};
struct LinTerm {  // int64[4]
    int64_t src_off, src_ld;
    double a_re, a_im;
};

template <bool CPLX>
__global__ __launch_bounds__(NT) void lincomb_kernel(const LinJob *__restrict__ jobs, const LinTerm *__restrict__ terms,
                                                     const double *__restrict__ src, double *__restrict__ dst) {
    const LinJob J = jobs[blockIdx.y];
    const int64_t total = J.rows * J.cols;
    const LinTerm *T = terms + J.term_begin;
    const int nt = (int)J.term_count;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < total; e += (int64_t)gridDim.x * NT) {
        const int64_t r = e / J.cols, c = e - r * J.cols;
        if (!CPLX) {
            double v = 0.0;
            for (int t = 0; t < nt; ++t) v = fma(T[t].a_re, src[T[t].src_off + r * T[t].src_ld + c], v);
            dst[J.dst_off + r * J.dst_ld + c] = v;
        } else {
            double2 v{0.0, 0.0};
            for (int t = 0; t < nt; ++t) {
                const double2 x = reinterpret_cast<const double2 *>(src)[T[t].src_off + r * T[t].src_ld + c];
                v.x += T[t].a_re * x.x - T[t].a_im * x.y;
                v.y += T[t].a_re * x.y + T[t].a_im * x.x;
            }
            reinterpret_cast<double2 *>(dst)[J.dst_off + r * J.dst_ld + c] = v;
        }
    }
}

struct ScaleJob {  // int64[6]
    int64_t x_off, pre, len, post, s_off, pad;
};

template <bool CPLX, bool SCPLX>
__global__ __launch_bounds__(NT) void scale_axis_kernel(const ScaleJob *__restrict__ jobs,
                                                        double *__restrict__ x,
                                                        const double *__restrict__ s) {
    const ScaleJob J = jobs[blockIdx.y];
    const int64_t total = J.pre * J.len * J.post;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < total; e += (int64_t)gridDim.x * NT) {
        const int64_t j = (e / J.post) % J.len;
        if (!CPLX) {
            x[J.x_off + e] *= s[J.s_off + j];
        } else {
            double2 v = reinterpret_cast<double2 *>(x)[J.x_off + e];
            if (SCPLX) {
                const double2 f = reinterpret_cast<const double2 *>(s)[J.s_off + j];
                v = double2{v.x * f.x - v.y * f.y, v.x * f.y + v.y * f.x};
            } else {
                const double f = s[J.s_off + j];
                v.x *= f;
                v.y *= f;
            }
            reinterpret_cast<double2 *>(x)[J.x_off + e] = v;
        }
    }
}

struct GatherJob {  // int64[8]
    int64_t dst_off, src_off, pre, len_src, len_dst, post, idx_off, pad;
};

template <bool CPLX>
__global__ __launch_bounds__(NT) void gather_axis_kernel(const GatherJob *__restrict__ jobs,
                                                         const int64_t *__restrict__ idx,
                                                         const double *__restrict__ src,
                                                         double *__restrict__ dst) {
    const GatherJob J = jobs[blockIdx.y];
    const int64_t total = J.pre * J.len_dst * J.post;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < total; e += (int64_t)gridDim.x * NT) {
        const int64_t l = e % J.post;
        const int64_t t = e / J.post;
        const int64_t j = t % J.len_dst;
        const int64_t i = t / J.len_dst;
        const int64_t so = J.src_off + (i * J.len_src + idx[J.idx_off + j]) * J.post + l;
        if (CPLX)
            reinterpret_cast<double2 *>(dst)[J.dst_off + e] = reinterpret_cast<const double2 *>(src)[so];
        else
            dst[J.dst_off + e] = src[so];
    }
}

// out[o_off + j] = sum_{i,l} |x[i, j, l]|^2 for blocks viewed as (pre, len, post): the per-slice norms that
// _qr_theta_Y0 takes with np.linalg.norm(block, axis=...) (truncation.py:452).  One wavefront per (job, j).
struct NormJob {  // int64[6]
    int64_t x_off, pre, len, post, o_off, pad;
};
template <bool CPLX>
__global__ __launch_bounds__(NT) void axis_sqnorm_kernel(const NormJob *__restrict__ jobs,
                                                         const int2 *__restrict__ rows,
                                                         const double *__restrict__ x, double *__restrict__ out) {
    const int gw = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int2 jr = rows[gw];
    if (jr.x < 0) return;
    const NormJob J = jobs[jr.x];
    const int64_t j = jr.y;
    double s = 0;
    const int64_t cnt = J.pre * J.post;
    for (int64_t e = lane; e < cnt; e += 64) {
        const int64_t i = e / J.post, l = e - i * J.post;
        const int64_t idx = J.x_off + (i * J.len + j) * J.post + l;
        if (CPLX) {
            const double2 v = reinterpret_cast<const double2 *>(x)[idx];
            s += v.x * v.x + v.y * v.y;
        } else {
            s = fma(x[idx], x[idx], s);
        }
    }
    s = wave_sum(s);
    if (lane == 0) out[J.o_off + j] = s;
}

// dtype conversion / conjugation over a flat arena.  MODE 0: f64->f64 copy, 1: f64->c128, 2: c128->f64 (real
// part), 3: c128->c128 (optionally conjugated)
template <int MODE>
__global__ __launch_bounds__(NT) void convert_kernel(int64_t n, const double *__restrict__ src,
                                                     double *__restrict__ dst, int conj) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        if (MODE == 0) {
            dst[i] = src[i];
        } else if (MODE == 1) {
            reinterpret_cast<double2 *>(dst)[i] = double2{src[i], 0.0};
        } else if (MODE == 2) {
            dst[i] = reinterpret_cast<const double2 *>(src)[i].x;
        } else {
            double2 v = reinterpret_cast<const double2 *>(src)[i];
            if (conj) v.y = -v.y;
            reinterpret_cast<double2 *>(dst)[i] = v;
        }
    }
}

inline int grid_x(int64_t max_elems) {
    int64_t g = (max_elems + NT * 4 - 1) / (NT * 4);
    if (g < 1) g = 1;
    if (g > 512) g = 512;
    return (int)g;
}
}  // namespace

extern "C" int tpa_copy_batch(int dtype, const int64_t *jobs_dev, int n_jobs, int64_t max_job_elems,
                              const void *src_base, void *dst_base, void *stream) {
    TPA_ARG_CHECK(dtype == TPA_F64 || dtype == TPA_C128);
    if (n_jobs <= 0) return 0;
    TPA_ARG_CHECK(n_jobs <= 65535);
    dim3 grid(grid_x(max_job_elems), n_jobs);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == TPA_F64)
        copy_batch_kernel<false><<<grid, NT, 0, st>>>((const CopyJob *)jobs_dev, (const double *)src_base, (double *)dst_base);
    else
        copy_batch_kernel<true><<<grid, NT, 0, st>>>((const CopyJob *)jobs_dev, (const double *)src_base, (double *)dst_base);
    TPA_LAUNCH_CHECK();
    return 0;
}

extern "C" int tpa_lincomb_batch(int dtype, const int64_t *jobs_dev, int n_jobs, const int64_t *terms_dev,
                                 int64_t max_job_elems, const void *src_base, void *dst_base, void *stream) {
    TPA_ARG_CHECK(dtype == TPA_F64 || dtype == TPA_C128);
    if (n_jobs <= 0) return 0;
    TPA_ARG_CHECK(n_jobs <= 65535);
    dim3 grid(grid_x(max_job_elems), n_jobs);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == TPA_F64)
        lincomb_kernel<false><<<grid, NT, 0, st>>>((const LinJob *)jobs_dev, (const LinTerm *)terms_dev, (const double *)src_base, (double *)dst_base);
    else
        lincomb_kernel<true><<<grid, NT, 0, st>>>((const LinJob *)jobs_dev, (const LinTerm *)terms_dev, (const double *)src_base, (double *)dst_base);
    TPA_LAUNCH_CHECK();
    return 0;
}

extern "C" int tpa_scale_axis_batch(int dtype, const int64_t *jobs_dev, int n_jobs,
                                    int64_t max_job_elems, void *x_base, const void *s_dev,
                                    int s_is_complex, void *stream) {
    TPA_ARG_CHECK(dtype == TPA_F64 || dtype == TPA_C128);
    TPA_ARG_CHECK(!(dtype == TPA_F64 && s_is_complex));
    if (n_jobs <= 0) return 0;
    TPA_ARG_CHECK(n_jobs <= 65535);
    dim3 grid(grid_x(max_job_elems), n_jobs);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == TPA_F64)
        scale_axis_kernel<false, false><<<grid, NT, 0, st>>>((const ScaleJob *)jobs_dev, (double *)x_base, (const double *)s_dev);
    else if (s_is_complex)
        scale_axis_kernel<true, true><<<grid, NT, 0, st>>>((const ScaleJob *)jobs_dev, (double *)x_base, (const double *)s_dev);
    else
        scale_axis_kernel<true, false><<<grid, NT, 0, st>>>((const ScaleJob *)jobs_dev, (double *)x_base, (const double *)s_dev);
    TPA_LAUNCH_CHECK();
    return 0;
}

// Ordered (Gram-Schmidt like) orthonormalisation step on the Gram matrix G = T T^H of row vectors sorted by DESCENDING weight:
// G <- strict lower triangle of G, diagonal (G_ii - 1) / 2, zero above.  Then T <- T - G T makes every vector orthogonal to the
// vectors BEFORE it (to second order in the defect) and moves no vector towards a later one: the clean-up of the small singular
// vectors of the block SVD (tenpy_amd/linalg/_svd_warm.py::ordered_rows).  jobs: int64[n][2] = {g_off, n}.
struct TriJob {
    int64_t g_off, n;
};

template <bool CPLX>
__global__ __launch_bounds__(NT) void tri_lower_kernel(const TriJob *__restrict__ jobs, double *__restrict__ g) {
    const TriJob J = jobs[blockIdx.y];
    const int64_t total = J.n * J.n;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < total; e += (int64_t)gridDim.x * NT) {
        const int64_t i = e / J.n, j = e - i * J.n;
        if (!CPLX) {
            double v = g[J.g_off + e];
            v = (i > j) ? v : (i == j) ? 0.5 * (v - 1.0) : 0.0;
            g[J.g_off + e] = v;
        } else {
            double2 v = reinterpret_cast<double2 *>(g)[J.g_off + e];
            v = (i > j) ? v : (i == j) ? double2{0.5 * (v.x - 1.0), 0.0} : double2{0.0, 0.0};
            reinterpret_cast<double2 *>(g)[J.g_off + e] = v;
        }
    }
}

extern "C" int tpa_tri_lower_batch(int dtype, const int64_t *jobs_dev, int n_jobs, int64_t max_job_elems, void *g_base,
                                   void *stream) {
    TPA_ARG_CHECK(dtype == TPA_F64 || dtype == TPA_C128);
    if (n_jobs <= 0) return 0;
    TPA_ARG_CHECK(n_jobs <= 65535);
    dim3 grid(grid_x(max_job_elems), n_jobs);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == TPA_F64)
        tri_lower_kernel<false><<<grid, NT, 0, st>>>((const TriJob *)jobs_dev, (double *)g_base);
    else
        tri_lower_kernel<true><<<grid, NT, 0, st>>>((const TriJob *)jobs_dev, (double *)g_base);
    TPA_LAUNCH_CHECK();
    return 0;
}

extern "C" int tpa_gather_axis_batch(int dtype, const int64_t *jobs_dev, int n_jobs,
                                     int64_t max_job_elems, const int64_t *idx_dev,
                                     const void *src_base, void *dst_base, void *stream) {
    TPA_ARG_CHECK(dtype == TPA_F64 || dtype == TPA_C128);
    if (n_jobs <= 0) return 0;
    TPA_ARG_CHECK(n_jobs <= 65535);
    dim3 grid(grid_x(max_job_elems), n_jobs);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == TPA_F64)
        gather_axis_kernel<false><<<grid, NT, 0, st>>>((const GatherJob *)jobs_dev, idx_dev, (const double *)src_base, (double *)dst_base);
    else
        gather_axis_kernel<true><<<grid, NT, 0, st>>>((const GatherJob *)jobs_dev, idx_dev, (const double *)src_base, (double *)dst_base);
    TPA_LAUNCH_CHECK();
    return 0;
}

extern "C" int tpa_convert(int from_dtype, int to_dtype, int64_t n, const void *src, void *dst, int conj,
                           void *stream) {
    TPA_ARG_CHECK((from_dtype == TPA_F64 || from_dtype == TPA_C128) && (to_dtype == TPA_F64 || to_dtype == TPA_C128));
    if (n <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    int g = grid_x(n) * 4;
    if (g > 2048) g = 2048;
    if (from_dtype == TPA_F64 && to_dtype == TPA_F64)
        convert_kernel<0><<<g, NT, 0, st>>>(n, (const double *)src, (double *)dst, 0);
    else if (from_dtype == TPA_F64)
        convert_kernel<1><<<g, NT, 0, st>>>(n, (const double *)src, (double *)dst, 0);
    else if (to_dtype == TPA_F64)
        convert_kernel<2><<<g, NT, 0, st>>>(n, (const double *)src, (double *)dst, 0);
    else
        convert_kernel<3><<<g, NT, 0, st>>>(n, (const double *)src, (double *)dst, conj);
    TPA_LAUNCH_CHECK();
    return 0;
}

extern "C" int tpa_axis_sqnorm_batch(int dtype, const int64_t *jobs_dev, const int32_t *rows_dev, int n_rows,
                                     const void *x_base, double *out_dev, void *stream) {
    TPA_ARG_CHECK(dtype == TPA_F64 || dtype == TPA_C128);
    if (n_rows <= 0) return 0;
    TPA_ARG_CHECK(n_rows % (NT / 64) == 0);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == TPA_F64)
        axis_sqnorm_kernel<false><<<n_rows / (NT / 64), NT, 0, st>>>((const NormJob *)jobs_dev, (const int2 *)rows_dev, (const double *)x_base, out_dev);
    else
        axis_sqnorm_kernel<true><<<n_rows / (NT / 64), NT, 0, st>>>((const NormJob *)jobs_dev, (const int2 *)rows_dev, (const double *)x_base, out_dev);
    TPA_LAUNCH_CHECK();
    return 0;
}
