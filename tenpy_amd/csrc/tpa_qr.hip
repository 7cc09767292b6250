// K6: batched Householder QR, one workgroup per charge block, gfx950.
//
// Replaces np.linalg.qr (LAPACK geqrf + orgqr) per block of the reference's npc.qr
// (np_conserved.py:4190).  A (m x n, row-major) -> Q (m x k), R (k x n), k = min(m, n), "reduced" mode.
// The factorisation runs in place on a row-major working copy R_full (m x n) held in the output /
// workspace; thread t owns trailing column(s) c = t, t+NT, ... so that the rank-1 update
//      A[j:, c] -= tau * v * (v^H A[j:, c])
// reads and writes rows coalesced across the workgroup (row-major blocks); the reflector v lives in
// LDS.  Q is then formed by applying the reflectors in reverse order to the first k columns of I.
#include "tpa_common.h"
#include <vector>

namespace {
constexpr int NT = 256;

struct QrJob {  // int64[8] device copy
    int64_t a_off, m, n, q_off, r_off, w_off, tau_off, pad;
};

template <bool CPLX>
struct Num;
template <>
struct Num<false> {
    using T = double;
    __device__ static T zero() { return 0.0; }
    __device__ static T conj(T a) { return a; }
    __device__ static T mul(T a, T b) { return a * b; }
    __device__ static T add(T a, T b) { return a + b; }
    __device__ static T sub(T a, T b) { return a - b; }
    __device__ static double abs2(T a) { return a * a; }
    __device__ static T scale(T a, double s) { return a * s; }
};
template <>
struct Num<true> {
    using T = double2;
    __device__ static T zero() { return double2{0, 0}; }
    __device__ static T conj(T a) { return double2{a.x, -a.y}; }
    __device__ static T mul(T a, T b) { return double2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
    __device__ static T add(T a, T b) { return double2{a.x + b.x, a.y + b.y}; }
    __device__ static T sub(T a, T b) { return double2{a.x - b.x, a.y - b.y}; }
    __device__ static double abs2(T a) { return a.x * a.x + a.y * a.y; }
    __device__ static T scale(T a, double s) { return double2{a.x * s, a.y * s}; }
};

// W: working copy (m x n row major).  v_j stored in LDS (length m - j, v[0] = 1 implicit handled
// explicitly), tau in global.  Reflector H = I - tau v v^H with H x = beta e1, beta real.
template <bool CPLX>
__global__ __launch_bounds__(NT) void qr_kernel(const QrJob *__restrict__ jobs,
                                                const typename Num<CPLX>::T *__restrict__ A,
                                                typename Num<CPLX>::T *__restrict__ Wb,
                                                typename Num<CPLX>::T *__restrict__ Vb,
                                                typename Num<CPLX>::T *__restrict__ Qb,
                                                typename Num<CPLX>::T *__restrict__ Rb) {
    using N = Num<CPLX>;
    using T = typename N::T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *v = reinterpret_cast<T *>(smem);  // m entries
    __shared__ double red[NT / 64];
    __shared__ T sh_tau;
    __shared__ int sh_active;
    const QrJob J = jobs[blockIdx.x];
    const int64_t m = J.m, n = J.n, k = (m < n) ? m : n;
    const int tid = threadIdx.x;
    T *W = Wb + J.w_off;    // m x n
    T *V = Vb + J.w_off;    // m x n storage for reflectors (column j holds v_j, rows j..m-1)
    T *tau = Vb + J.tau_off;  // k entries
    const T *Ain = A + J.a_off;
    for (int64_t e = tid; e < m * n; e += NT) W[e] = Ain[e];
    __syncthreads();
    for (int64_t j = 0; j < k; ++j) {
        // --- build reflector from x = W[j:m, j]
        double s2 = 0;
        for (int64_t i = j + 1 + tid; i < m; i += NT) s2 += N::abs2(W[i * n + j]);
        s2 = block_sum<NT>(s2, red);
        if (tid == 0) {
            const T x0 = W[j * n + j];
            const double a0 = N::abs2(x0);
            const double xnorm = sqrt(a0 + s2);
            T t;
            T vscale;  // 1/(x0 - beta)
            double beta;
            if (s2 == 0.0 && (!CPLX || (CPLX && N::abs2(N::sub(x0, N::conj(x0))) == 0.0))) {
                // already upper triangular in this column (and real diagonal): H = I
                t = N::zero();
                vscale = N::zero();
                beta = 0;  // unused
                sh_tau = t;
                v[j] = N::zero();
                tau[j] = t;
                sh_active = 0;
            } else {
                double re0;
                if (CPLX)
                    re0 = reinterpret_cast<const double *>(&x0)[0];
                else
                    re0 = *reinterpret_cast<const double *>(&x0);
                beta = (re0 >= 0) ? -xnorm : xnorm;
                // tau = (beta - x0)/beta ; v = x/(x0 - beta), v0 = 1
                T bmx;  // beta - x0
                T xmb;  // x0 - beta
                if (CPLX) {
                    double2 x0c = *reinterpret_cast<const double2 *>(&x0);
                    double2 b1{beta - x0c.x, -x0c.y};
                    double2 b2{x0c.x - beta, x0c.y};
                    bmx = *reinterpret_cast<T *>(&b1);
                    xmb = *reinterpret_cast<T *>(&b2);
                } else {
                    double b1 = beta - re0, b2 = re0 - beta;
                    bmx = *reinterpret_cast<T *>(&b1);
                    xmb = *reinterpret_cast<T *>(&b2);
                }
                t = N::scale(bmx, 1.0 / beta);
                const double d = N::abs2(xmb);
                vscale = N::scale(N::conj(xmb), 1.0 / d);
                sh_tau = t;
                tau[j] = t;
                // store: W[j,j] = beta ; v0 = 1
                T bb;
                if (CPLX) {
                    double2 tmp{beta, 0};
                    bb = *reinterpret_cast<T *>(&tmp);
                } else {
                    bb = *reinterpret_cast<T *>(&beta);
                }
                W[j * n + j] = bb;
                v[j] = vscale;  // temporarily pass the scale through LDS slot j
                sh_active = 1;
            }
        }
        __syncthreads();
        const bool active = (sh_active != 0);
        const T vs = v[j];
        __syncthreads();
        if (active) {
            for (int64_t i = j + 1 + tid; i < m; i += NT) {
                const T vi = N::mul(W[i * n + j], vs);
                v[i] = vi;
                V[i * n + j] = vi;
                W[i * n + j] = N::zero();
            }
            if (tid == 0) {
                T one;
                if (CPLX) {
                    double2 tmp{1, 0};
                    one = *reinterpret_cast<T *>(&tmp);
                } else {
                    double tmp = 1;
                    one = *reinterpret_cast<T *>(&tmp);
                }
                v[j] = one;
                V[j * n + j] = one;
            }
        } else {
            for (int64_t i = j + tid; i < m; i += NT) V[i * n + j] = N::zero();
        }
        __syncthreads();
        if (active) {
            // --- apply H^H = I - conj(tau) v v^H to trailing columns (so that H^H A = R; A = H R)
            const T tc = N::conj(sh_tau);
            for (int64_t c = j + 1 + tid; c < n; c += NT) {
                T w = N::zero();
                for (int64_t i = j; i < m; ++i) w = N::add(w, N::mul(N::conj(v[i]), W[i * n + c]));
                w = N::mul(tc, w);
                for (int64_t i = j; i < m; ++i) W[i * n + c] = N::sub(W[i * n + c], N::mul(v[i], w));
            }
        }
        __syncthreads();
    }
    // --- R = first k rows of W
    T *R = Rb + J.r_off;
    for (int64_t e = tid; e < k * n; e += NT) R[e] = W[e];
    // --- Q = H_0 H_1 ... H_{k-1} [I_k; 0]  (m x k): apply in reverse order; reuse W as scratch? no:
    // write directly into Q (m x k, row-major)
    T *Q = Qb + J.q_off;
    for (int64_t e = tid; e < m * k; e += NT) {
        const int64_t i = e / k, c = e % k;
        T val = N::zero();
        if (i == c) {
            if (CPLX) {
                double2 tmp{1, 0};
                val = *reinterpret_cast<T *>(&tmp);
            } else {
                double tmp = 1;
                val = *reinterpret_cast<T *>(&tmp);
            }
        }
        Q[e] = val;
    }
    __syncthreads();
    for (int64_t j = k - 1; j >= 0; --j) {
        for (int64_t i = j + tid; i < m; i += NT) v[i] = V[i * n + j];
        __syncthreads();
        const T t = tau[j];
        if (N::abs2(t) != 0.0) {
            // Q[j:, c] -= tau v (v^H Q[j:, c]) for c >= j (columns < j are still unit vectors e_c, c<j, untouched)
            for (int64_t c = j + tid; c < k; c += NT) {
                T w = N::zero();
                for (int64_t i = j; i < m; ++i) w = N::add(w, N::mul(N::conj(v[i]), Q[i * k + c]));
                w = N::mul(t, w);
                for (int64_t i = j; i < m; ++i) Q[i * k + c] = N::sub(Q[i * k + c], N::mul(v[i], w));
            }
        }
        __syncthreads();
    }
}
}  // namespace

int tpa_qr_wy_internal(int dtype, const int64_t *jobs_host, int n_jobs, const void *a_base, void *q_base, void *r_base,
                       void *stream);
int tpa_qr_use_wy = 1;   // test hook (tpa_qr_set_algorithm): 0 = always the one-workgroup kernel

int tpa_qr_lookahead = 1;      // one launch per panel (real data, tpa_qr_la.inc in tpa_svd.hip); bit 1 of tpa_qr_set_algorithm = the two-kernel path of rounds 2-4

extern "C" int tpa_qr_set_algorithm(int v) {
    tpa_qr_use_wy = (v & 1) ? 0 : 1;
    tpa_qr_lookahead = (v & 2) ? 0 : 1;
    return 0;
}

// workspace requirement: per job 2*m*n + k elements (W, V, tau); allocated internally via hipMallocAsync
extern "C" int tpa_qr_batch(int dtype, const int64_t *jobs_host, int n_jobs, const void *a_base,
                            void *q_base, void *r_base, void *stream) {
    TPA_ARG_CHECK(dtype == TPA_F64 || dtype == TPA_C128);
    if (n_jobs <= 0) return 0;
    {   // The work areas come from hipMallocAsync / hipFreeAsync.  The default pool gives freed memory back to the OS at the next
        // synchronisation (release threshold 0), so every call would map its ~100 MB afresh: keep up to 8 GB of what the pool has
        // handed out (bounded -- ADVICE r5: an unbounded threshold kept the multi-GB peak of a batched TEBD QR for the life of the
        // process, invisible to torch's allocator and to the memory budgets of linalg/_device.py).
        static bool pool_kept = false;
        if (!pool_kept) {
            int devid = 0;
            hipMemPool_t pool;
            if (hipGetDevice(&devid) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, devid) == hipSuccess) {
                uint64_t thr = (uint64_t)8 << 30;
                (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
            }
            (void)hipGetLastError();
            pool_kept = true;
        }
    }
    {   // large blocks: blocked compact-WY QR on the matrix cores (tpa_svd.hip); this file's one-workgroup kernel is
        // launch-cheaper for small blocks but streams the whole trailing matrix through ONE CU per column
        int64_t kmax = 0, dmax = 0;
        for (int b = 0; b < n_jobs; ++b) {
            TPA_ARG_CHECK(jobs_host[8 * b + 1] > 0 && jobs_host[8 * b + 2] > 0);
            kmax = std::max(kmax, std::min(jobs_host[8 * b + 1], jobs_host[8 * b + 2]));
            dmax = std::max(dmax, jobs_host[8 * b + 1]);
        }
        if (tpa_qr_use_wy && kmax >= 32 && dmax <= ((dtype == TPA_F64) ? 8192 : 2048))
            return tpa_qr_wy_internal(dtype, jobs_host, n_jobs, a_base, q_base, r_base, stream);
    }
    hipStream_t st = (hipStream_t)stream;
    const int64_t esz = (dtype == TPA_C128) ? 16 : 8;
    std::vector<QrJob> jobs(n_jobs);
    int64_t w_elems = 0, mmax = 0;
    for (int b = 0; b < n_jobs; ++b) {
        const int64_t *j = jobs_host + 8 * b;
        QrJob &J = jobs[b];
        J.a_off = j[0];
        J.m = j[1];
        J.n = j[2];
        J.q_off = j[3];
        J.r_off = j[4];
        TPA_ARG_CHECK(J.m > 0 && J.n > 0);
        J.w_off = w_elems;
        w_elems += J.m * J.n;
        mmax = std::max(mmax, J.m);
    }
    int64_t tau_base = w_elems;  // tau stored after the V region (same buffer as V)
    for (int b = 0; b < n_jobs; ++b) {
        jobs[b].tau_off = tau_base;
        tau_base += std::min(jobs[b].m, jobs[b].n);
    }
    const size_t lds = (size_t)mmax * esz;
    TPA_ARG_CHECK(lds <= 150 * 1024);
    char *buf = nullptr;
    const size_t wbytes = (size_t)w_elems * esz, vbytes = (size_t)tau_base * esz;
    const size_t jbytes = jobs.size() * sizeof(QrJob);
    const size_t total = ((wbytes + 255) / 256 + (vbytes + 255) / 256 + (jbytes + 255) / 256) * 256;
    TPA_HIP_CHECK(hipMallocAsync((void **)&buf, total, st));
    char *Wb = buf, *Vb = buf + (wbytes + 255) / 256 * 256, *Jb = Vb + (vbytes + 255) / 256 * 256;
    TPA_HIP_CHECK(hipMemcpyAsync(Jb, jobs.data(), jbytes, hipMemcpyHostToDevice, st));
    if (dtype == TPA_F64) {
        TPA_HIP_CHECK(hipFuncSetAttribute((const void *)qr_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        qr_kernel<false><<<n_jobs, NT, lds, st>>>((const QrJob *)Jb, (const double *)a_base, (double *)Wb, (double *)Vb, (double *)q_base, (double *)r_base);
    } else {
        TPA_HIP_CHECK(hipFuncSetAttribute((const void *)qr_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        qr_kernel<true><<<n_jobs, NT, lds, st>>>((const QrJob *)Jb, (const double2 *)a_base, (double2 *)Wb, (double2 *)Vb, (double2 *)q_base, (double2 *)r_base);
    }
    TPA_LAUNCH_CHECK();
    TPA_HIP_CHECK(hipStreamSynchronize(st));  // `jobs` (pageable) must outlive its async upload
    TPA_HIP_CHECK(hipFreeAsync(buf, st));
    return 0;
}
