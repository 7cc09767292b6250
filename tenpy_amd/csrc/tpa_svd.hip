// K5: batched block SVD by one-sided (Hestenes) Jacobi with wavefront-level reductions, gfx950.
//
// Replaces the per-charge-block LAPACK call of the reference (np_conserved.py:4970-4980 `svd_flat`
// -> svd_robust.py:36-75 gesdd/gesvd).  All charge blocks of one npc.svd are processed together.
//
// For a block A (m x n) we orthogonalise the ROWS of W (R x L, R = min(m,n) <= L = max(m,n)):
//   m <  n : W = A            ->  A = G^H Sigma Y          U = G^H,   VH = Y
//   m >= n : W = A^T          ->  A = Y^T Sigma conj(G)    U = Y^T,   VH = conj(G)
// where G (R x R, starts as identity) accumulates the plane rotations applied to the rows, and
// Y = Sigma^-1 W_final has orthonormal rows.  One wavefront owns one row pair of a round-robin
// tournament round: pass 1 reduces (|x|^2, |y|^2, x.conj(y)) with __shfl_xor, pass 2 applies the
// rotation to the two rows of W and of G.  A round = one launch over all pairs of all blocks; a sweep
// = (Rmax_even - 1) rounds; the host tests a device-side rotation counter once per sweep.
// One-sided Jacobi computes small singular values to high *relative* accuracy (better than gesdd),
// which is what the 1e-10 parity bound on singular values needs.
#include "tpa_common.h"
#include <algorithm>
#include <cmath>
#include <numeric>
#include <vector>

namespace {

constexpr int NT = 256;  // 4 wavefronts = 4 row pairs per workgroup

struct SvdJob {  // int64[12], device copy
    int64_t w_off, g_off, R, L, Rpad, a_off, m, n, u_off, s_off, vh_off, sig_off;
};

template <bool CPLX>
__global__ __launch_bounds__(NT) void svd_init_kernel(const SvdJob *__restrict__ jobs,
                                                      const int2 *__restrict__ rows,
                                                      const double *__restrict__ A,
                                                      double *__restrict__ W, double *__restrict__ G) {
    // one wavefront per row of W (and of G)
    const int gw = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int2 jr = rows[gw];
    if (jr.x < 0) return;
    const SvdJob J = jobs[jr.x];
    const int64_t r = jr.y;
    const bool tr = (J.m >= J.n);
    for (int64_t c = lane; c < J.L; c += 64) {
        const int64_t src = J.a_off + (tr ? (c * J.n + r) : (r * J.n + c));
        if (CPLX)
            reinterpret_cast<double2 *>(W)[J.w_off + r * J.L + c] = reinterpret_cast<const double2 *>(A)[src];
        else
            W[J.w_off + r * J.L + c] = A[src];
    }
    for (int64_t c = lane; c < J.R; c += 64) {
        if (CPLX)
            reinterpret_cast<double2 *>(G)[J.g_off + r * J.R + c] = double2{(c == r) ? 1.0 : 0.0, 0.0};
        else
            G[J.g_off + r * J.R + c] = (c == r) ? 1.0 : 0.0;
    }
}

template <bool CPLX>
__global__ __launch_bounds__(NT) void svd_round_kernel(const SvdJob *__restrict__ jobs,
                                                       const int2 *__restrict__ pairs, int round,
                                                       double *__restrict__ W, double *__restrict__ G,
                                                       unsigned int *__restrict__ n_rot) {
    const int gw = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int2 jp = pairs[gw];
    if (jp.x < 0) return;
    const SvdJob J = jobs[jp.x];
    const int64_t np = J.Rpad;  // even number of players
    if (np < 2) return;
    const int64_t mod = np - 1;
    const int64_t r = round % mod;
    const int64_t i = jp.y;
    int64_t p, q;
    if (i == 0) {
        p = np - 1;
        q = r;
    } else {
        p = (r + i) % mod;
        q = (r - i + mod) % mod;
    }
    if (p > q) {
        const int64_t t = p;
        p = q;
        q = t;
    }
    if (q >= J.R) return;  // bye
    const int64_t L = J.L, R = J.R;
    double *x = W + (CPLX ? 2 : 1) * (J.w_off + p * L);
    double *y = W + (CPLX ? 2 : 1) * (J.w_off + q * L);
    double a = 0, b = 0, gr = 0, gi = 0;
    if (!CPLX) {
        for (int64_t c = lane; c < L; c += 64) {
            const double xv = x[c], yv = y[c];
            a = fma(xv, xv, a);
            b = fma(yv, yv, b);
            gr = fma(xv, yv, gr);
        }
    } else {
        for (int64_t c = lane; c < L; c += 64) {
            const double2 xv = reinterpret_cast<double2 *>(x)[c], yv = reinterpret_cast<double2 *>(y)[c];
            a += xv.x * xv.x + xv.y * xv.y;
            b += yv.x * yv.x + yv.y * yv.y;
            gr += xv.x * yv.x + xv.y * yv.y;  // x * conj(y)
            gi += xv.y * yv.x - xv.x * yv.y;
        }
    }
    a = wave_sum(a);
    b = wave_sum(b);
    gr = wave_sum(gr);
    if (CPLX) gi = wave_sum(gi);
    const double g2 = gr * gr + gi * gi;
    const double tol = 2.220446049250313e-16 * sqrt((double)L);
    if (!(g2 > tol * tol * a * b) || a == 0.0 || b == 0.0) return;  // already orthogonal (or NaN)
    const double gabs = sqrt(g2);
    const double zeta = (b - a) / (2.0 * gabs);
    const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double c = 1.0 / sqrt(1.0 + t * t);
    const double s = c * t;
    const double pr = gr / gabs, pi = gi / gabs;  // e^{i phi}
    // x' = c x - s e^{i phi} y ;  y' = s e^{-i phi} x + c y
    if (!CPLX) {
        const double sp = s * pr;
        for (int64_t cc = lane; cc < L; cc += 64) {
            const double xv = x[cc], yv = y[cc];
            x[cc] = c * xv - sp * yv;
            y[cc] = sp * xv + c * yv;
        }
        double *gx = G + J.g_off + p * R, *gy = G + J.g_off + q * R;
        for (int64_t cc = lane; cc < R; cc += 64) {
            const double xv = gx[cc], yv = gy[cc];
            gx[cc] = c * xv - sp * yv;
            gy[cc] = sp * xv + c * yv;
        }
    } else {
        const double sr = s * pr, si = s * pi;
        auto rot = [&](double2 *xx, double2 *yy, int64_t len) {
            for (int64_t cc = lane; cc < len; cc += 64) {
                const double2 xv = xx[cc], yv = yy[cc];
                double2 xn, yn;
                xn.x = c * xv.x - (sr * yv.x - si * yv.y);
                xn.y = c * xv.y - (sr * yv.y + si * yv.x);
                yn.x = (sr * xv.x + si * xv.y) + c * yv.x;  // (sr - i si) * x
                yn.y = (sr * xv.y - si * xv.x) + c * yv.y;
                xx[cc] = xn;
                yy[cc] = yn;
            }
        };
        rot(reinterpret_cast<double2 *>(x), reinterpret_cast<double2 *>(y), L);
        rot(reinterpret_cast<double2 *>(G) + J.g_off + p * R, reinterpret_cast<double2 *>(G) + J.g_off + q * R, R);
    }
    if (lane == 0) atomicAdd(n_rot, 1u);
}

template <bool CPLX>
__global__ __launch_bounds__(NT) void svd_norms_kernel(const SvdJob *__restrict__ jobs,
                                                       const int2 *__restrict__ rows,
                                                       const double *__restrict__ W,
                                                       double *__restrict__ sig) {
    const int gw = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int2 jr = rows[gw];
    if (jr.x < 0) return;
    const SvdJob J = jobs[jr.x];
    const double *x = W + (CPLX ? 2 : 1) * (J.w_off + (int64_t)jr.y * J.L);
    const int64_t len = (CPLX ? 2 : 1) * J.L;
    double a = 0;
    for (int64_t c = lane; c < len; c += 64) a = fma(x[c], x[c], a);
    a = wave_sum(a);
    if (lane == 0) sig[J.sig_off + jr.y] = sqrt(a);
}

// rows[gw] = (job, sorted position jj); perm[sig_off + jj] = source row
template <bool CPLX>
__global__ __launch_bounds__(NT) void svd_finish_kernel(const SvdJob *__restrict__ jobs,
                                                        const int2 *__restrict__ rows,
                                                        const int64_t *__restrict__ perm,
                                                        const double *__restrict__ W,
                                                        const double *__restrict__ G,
                                                        const double *__restrict__ sig,
                                                        double *__restrict__ U, double *__restrict__ S,
                                                        double *__restrict__ VH) {
    const int gw = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int2 jr = rows[gw];
    if (jr.x < 0) return;
    const SvdJob J = jobs[jr.x];
    const int64_t jj = jr.y;
    const int64_t j = perm[J.sig_off + jj];
    const double sg = sig[J.sig_off + j];
    const double inv = (sg > 0.0) ? 1.0 / sg : 0.0;
    const bool tr = (J.m >= J.n);
    const int64_t k = J.R;
    if (lane == 0) S[J.s_off + jj] = sg;
    // Y row j = W[j,:]/sigma  -> VH row (m<n) or U column (m>=n)
    for (int64_t c = lane; c < J.L; c += 64) {
        const int64_t dst = tr ? (J.u_off + c * k + jj) : (J.vh_off + jj * J.n + c);
        double *D = tr ? U : VH;
        if (CPLX) {
            double2 v = reinterpret_cast<const double2 *>(W)[J.w_off + j * J.L + c];
            reinterpret_cast<double2 *>(D)[dst] = double2{v.x * inv, v.y * inv};
        } else {
            D[dst] = W[J.w_off + j * J.L + c] * inv;
        }
    }
    // conj(G row j) -> U column (m<n: U = G^H) or VH row (m>=n: VH = conj(G))
    for (int64_t c = lane; c < J.R; c += 64) {
        const int64_t dst = tr ? (J.vh_off + jj * J.n + c) : (J.u_off + c * k + jj);
        double *D = tr ? VH : U;
        if (CPLX) {
            double2 v = reinterpret_cast<const double2 *>(G)[J.g_off + j * J.R + c];
            reinterpret_cast<double2 *>(D)[dst] = double2{v.x, -v.y};
        } else {
            D[dst] = G[J.g_off + j * J.R + c];
        }
    }
}

struct Layout {
    std::vector<SvdJob> jobs;
    std::vector<int2> rows;   // (job,row) per wavefront, padded to multiple of 4 with (-1,-1)
    std::vector<int2> pairs;  // (job,pair)
    int64_t w_elems = 0, g_elems = 0, sig_elems = 0, rmax_pad = 0;
    // byte offsets inside work buffer
    int64_t off_w = 0, off_g = 0, off_sig = 0, off_perm = 0, off_jobs = 0, off_rows = 0, off_pairs = 0,
            off_cnt = 0, total = 0;
};

inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

Layout make_layout(int dtype, const int64_t *jobs_host, int n_jobs) {
    Layout lay;
    const int64_t esz = (dtype == TPA_C128) ? 16 : 8;
    for (int b = 0; b < n_jobs; ++b) {
        const int64_t *j = jobs_host + 8 * b;
        SvdJob J;
        J.a_off = j[0];
        J.m = j[1];
        J.n = j[2];
        J.u_off = j[3];
        J.s_off = j[4];
        J.vh_off = j[5];
        J.R = std::min(J.m, J.n);
        J.L = std::max(J.m, J.n);
        J.Rpad = (J.R + 1) / 2 * 2;
        J.w_off = lay.w_elems;
        J.g_off = lay.g_elems;
        J.sig_off = lay.sig_elems;
        lay.w_elems += J.R * J.L;
        lay.g_elems += J.R * J.R;
        lay.sig_elems += J.R;
        lay.rmax_pad = std::max(lay.rmax_pad, J.Rpad);
        for (int64_t r = 0; r < J.R; ++r) lay.rows.push_back(int2{b, (int)r});
        for (int64_t p = 0; p < J.Rpad / 2; ++p) lay.pairs.push_back(int2{b, (int)p});
        lay.jobs.push_back(J);
    }
    while (lay.rows.size() % (NT / 64)) lay.rows.push_back(int2{-1, -1});
    while (lay.pairs.size() % (NT / 64)) lay.pairs.push_back(int2{-1, -1});
    int64_t o = 0;
    lay.off_w = o;
    o = align_up(o + lay.w_elems * esz, 256);
    lay.off_g = o;
    o = align_up(o + lay.g_elems * esz, 256);
    lay.off_sig = o;
    o = align_up(o + lay.sig_elems * 8, 256);
    lay.off_perm = o;
    o = align_up(o + lay.sig_elems * 8, 256);
    lay.off_jobs = o;
    o = align_up(o + (int64_t)lay.jobs.size() * sizeof(SvdJob), 256);
    lay.off_rows = o;
    o = align_up(o + (int64_t)lay.rows.size() * sizeof(int2), 256);
    lay.off_pairs = o;
    o = align_up(o + (int64_t)lay.pairs.size() * sizeof(int2), 256);
    lay.off_cnt = o;
    o = align_up(o + 256, 256);
    lay.total = o;
    return lay;
}

template <bool CPLX>
int svd_run(const Layout &lay, int n_jobs, const void *a_base, void *u_base, double *s_dev,
            void *vh_base, char *work, int max_sweeps, int *sweeps_done, hipStream_t st) {
    double *W = (double *)(work + lay.off_w);
    double *G = (double *)(work + lay.off_g);
    double *sig = (double *)(work + lay.off_sig);
    int64_t *perm = (int64_t *)(work + lay.off_perm);
    SvdJob *jobs = (SvdJob *)(work + lay.off_jobs);
    int2 *rows = (int2 *)(work + lay.off_rows);
    int2 *pairs = (int2 *)(work + lay.off_pairs);
    unsigned int *cnt = (unsigned int *)(work + lay.off_cnt);
    TPA_HIP_CHECK(hipMemcpyAsync(jobs, lay.jobs.data(), lay.jobs.size() * sizeof(SvdJob), hipMemcpyHostToDevice, st));
    TPA_HIP_CHECK(hipMemcpyAsync(rows, lay.rows.data(), lay.rows.size() * sizeof(int2), hipMemcpyHostToDevice, st));
    TPA_HIP_CHECK(hipMemcpyAsync(pairs, lay.pairs.data(), lay.pairs.size() * sizeof(int2), hipMemcpyHostToDevice, st));
    // pageable host memory: the copies above are staged before returning, vectors may die later.
    const int g_rows = (int)(lay.rows.size() / (NT / 64));
    const int g_pairs = (int)(lay.pairs.size() / (NT / 64));
    if (g_rows == 0) {
        if (sweeps_done) *sweeps_done = 0;
        return 0;
    }
    svd_init_kernel<CPLX><<<g_rows, NT, 0, st>>>(jobs, rows, (const double *)a_base, W, G);
    TPA_LAUNCH_CHECK();
    int sweep = 0;
    bool converged = (lay.rmax_pad < 2);
    const int rounds = (int)std::max<int64_t>(lay.rmax_pad - 1, 1);
    while (!converged && sweep < max_sweeps) {
        TPA_HIP_CHECK(hipMemsetAsync(cnt, 0, sizeof(unsigned int), st));
        for (int r = 0; r < rounds; ++r) {
            svd_round_kernel<CPLX><<<g_pairs, NT, 0, st>>>(jobs, pairs, r, W, G, cnt);
        }
        TPA_LAUNCH_CHECK();
        unsigned int h = 0;
        TPA_HIP_CHECK(hipMemcpyAsync(&h, cnt, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
        TPA_HIP_CHECK(hipStreamSynchronize(st));
        ++sweep;
        converged = (h == 0);
    }
    if (sweeps_done) *sweeps_done = sweep;
    svd_norms_kernel<CPLX><<<g_rows, NT, 0, st>>>(jobs, rows, W, sig);
    TPA_LAUNCH_CHECK();
    std::vector<double> hs(lay.sig_elems);
    TPA_HIP_CHECK(hipMemcpyAsync(hs.data(), sig, lay.sig_elems * 8, hipMemcpyDeviceToHost, st));
    TPA_HIP_CHECK(hipStreamSynchronize(st));
    std::vector<int64_t> hp(lay.sig_elems);
    bool bad = false;
    for (int b = 0; b < n_jobs; ++b) {
        const SvdJob &J = lay.jobs[b];
        int64_t *p = hp.data() + J.sig_off;
        const double *s = hs.data() + J.sig_off;
        std::iota(p, p + J.R, (int64_t)0);
        std::stable_sort(p, p + J.R, [s](int64_t x, int64_t y) { return s[x] > s[y]; });
        for (int64_t i = 0; i < J.R; ++i)
            if (!std::isfinite(s[i])) bad = true;
    }
    if (bad) {
        snprintf(tpa_errbuf, sizeof(tpa_errbuf), "tpa_svd_batch: NaN/Inf in singular values");
        return TPA_E_NAN;
    }
    TPA_HIP_CHECK(hipMemcpyAsync(perm, hp.data(), lay.sig_elems * 8, hipMemcpyHostToDevice, st));
    svd_finish_kernel<CPLX><<<g_rows, NT, 0, st>>>(jobs, rows, perm, W, G, sig, (double *)u_base, s_dev, (double *)vh_base);
    TPA_LAUNCH_CHECK();
    TPA_HIP_CHECK(hipStreamSynchronize(st));  // hp must outlive the async copy
    if (!converged) {
        snprintf(tpa_errbuf, sizeof(tpa_errbuf), "tpa_svd_batch: no convergence in %d sweeps", max_sweeps);
        return TPA_E_NOCONV;
    }
    return 0;
}

}  // namespace

extern "C" int64_t tpa_svd_worksize(int dtype, const int64_t *jobs_host, int n_jobs) {
    if (n_jobs <= 0) return 256;
    return make_layout(dtype, jobs_host, n_jobs).total;
}

extern "C" int tpa_svd_batch(int dtype, const int64_t *jobs_host, int n_jobs, const void *a_base,
                             void *u_base, double *s_dev, void *vh_base, void *work_dev,
                             int64_t work_bytes, int max_sweeps, double tol, int *sweeps_done,
                             void *stream) {
    (void)tol;
    TPA_ARG_CHECK(dtype == TPA_F64 || dtype == TPA_C128);
    if (n_jobs <= 0) return 0;
    for (int b = 0; b < n_jobs; ++b) TPA_ARG_CHECK(jobs_host[8 * b + 1] > 0 && jobs_host[8 * b + 2] > 0);
    Layout lay = make_layout(dtype, jobs_host, n_jobs);
    TPA_ARG_CHECK(work_bytes >= lay.total);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == TPA_F64)
        return svd_run<false>(lay, n_jobs, a_base, u_base, s_dev, vh_base, (char *)work_dev, max_sweeps, sweeps_done, st);
    return svd_run<true>(lay, n_jobs, a_base, u_base, s_dev, vh_base, (char *)work_dev, max_sweeps, sweeps_done, st);
}

// ================================================================================================
// K7: batched Hermitian eigendecomposition on the same Jacobi machinery.
// Replaces np.linalg.eigh per block (np_conserved.py:5059-5061).  A Hermitian block is shifted to be
// positive definite, A' = A + mu I with mu = 2 ||A||_F >= 2 rho(A); then the SVD A' = U S U^H *is* its
// eigendecomposition (no +/-lambda mixing of singular subspaces), and lambda_j = S_j - mu.
// Absolute accuracy ~ eps * ||A||_F, the same class as LAPACK's eigh.
namespace {

struct EighJob {  // int64[8]
    int64_t a_off, n, w_off, v_off, ap_off, s_off, pad0, pad1;
};

template <bool CPLX>
__global__ __launch_bounds__(NT) void eigh_shift_kernel(const EighJob *__restrict__ jobs,
                                                        const double *__restrict__ A,
                                                        double *__restrict__ Ap, double *__restrict__ mu) {
    __shared__ double red[NT / 64];
    const EighJob J = jobs[blockIdx.x];
    const int64_t n = J.n, tot = n * n * (CPLX ? 2 : 1);
    const double *a = A + (CPLX ? 2 : 1) * J.a_off;
    double *ap = Ap + (CPLX ? 2 : 1) * J.ap_off;
    double s = 0;
    for (int64_t e = threadIdx.x; e < tot; e += NT) s = fma(a[e], a[e], s);
    s = block_sum<NT>(s, red);
    // mu = 2 ||A||_F (1 if A == 0): spectrum of A' lies in [||A||_F, 3 ||A||_F] > 0, so every singular
    // vector is well defined and the Jacobi iteration sees a condition number <= 3.
    const double m = (s > 0.0) ? 2.0 * sqrt(s) : 1.0;
    if (threadIdx.x == 0) mu[blockIdx.x] = m;
    for (int64_t e = threadIdx.x; e < n * n; e += NT) {
        const int64_t i = e / n, j = e % n;
        if (CPLX) {
            // enforce Hermitian symmetry from the lower triangle like UPLO='L'
            double2 v = (i >= j) ? reinterpret_cast<const double2 *>(a)[i * n + j]
                                 : reinterpret_cast<const double2 *>(a)[j * n + i];
            if (i < j) v.y = -v.y;
            if (i == j) {
                v.x += m;
                v.y = 0;
            }
            reinterpret_cast<double2 *>(ap)[e] = v;
        } else {
            double v = (i >= j) ? a[i * n + j] : a[j * n + i];
            if (i == j) v += m;
            ap[e] = v;
        }
    }
}

template <bool CPLX>
__global__ __launch_bounds__(NT) void eigh_finish_kernel(const EighJob *__restrict__ jobs,
                                                         const double *__restrict__ U,
                                                         const double *__restrict__ S,
                                                         const double *__restrict__ mu,
                                                         double *__restrict__ Wout, double *__restrict__ V) {
    const EighJob J = jobs[blockIdx.x];
    const int64_t n = J.n;
    const double m = mu[blockIdx.x];
    for (int64_t j = threadIdx.x; j < n; j += NT) Wout[J.w_off + j] = S[J.s_off + (n - 1 - j)] - m;
    for (int64_t e = threadIdx.x; e < n * n; e += NT) {
        const int64_t i = e / n, j = e % n;
        if (CPLX)
            reinterpret_cast<double2 *>(V)[J.v_off + e] = reinterpret_cast<const double2 *>(U)[J.ap_off + i * n + (n - 1 - j)];
        else
            V[J.v_off + e] = U[J.ap_off + i * n + (n - 1 - j)];
    }
}

struct EighLayout {
    std::vector<EighJob> jobs;
    std::vector<int64_t> svd_jobs;  // int64[8] per job, offsets into the workspace planes
    int64_t mat_elems = 0, s_elems = 0;
    int64_t off_ap = 0, off_u = 0, off_vh = 0, off_s = 0, off_mu = 0, off_jobs = 0, off_svd = 0, total = 0;
};

EighLayout make_eigh_layout(int dtype, const int64_t *jobs_host, int n_jobs) {
    EighLayout lay;
    const int64_t esz = (dtype == TPA_C128) ? 16 : 8;
    for (int b = 0; b < n_jobs; ++b) {
        const int64_t *j = jobs_host + 8 * b;
        EighJob J{};
        J.a_off = j[0];
        J.n = j[1];
        J.w_off = j[2];
        J.v_off = j[3];
        J.ap_off = lay.mat_elems;
        J.s_off = lay.s_elems;
        lay.mat_elems += J.n * J.n;
        lay.s_elems += J.n;
        lay.jobs.push_back(J);
        const int64_t sj[8] = {J.ap_off, J.n, J.n, J.ap_off, J.s_off, J.ap_off, 0, 0};
        lay.svd_jobs.insert(lay.svd_jobs.end(), sj, sj + 8);
    }
    int64_t o = 0;
    lay.off_ap = o;
    o = align_up(o + lay.mat_elems * esz, 256);
    lay.off_u = o;
    o = align_up(o + lay.mat_elems * esz, 256);
    lay.off_vh = o;
    o = align_up(o + lay.mat_elems * esz, 256);
    lay.off_s = o;
    o = align_up(o + lay.s_elems * 8, 256);
    lay.off_mu = o;
    o = align_up(o + (int64_t)n_jobs * 8, 256);
    lay.off_jobs = o;
    o = align_up(o + (int64_t)n_jobs * sizeof(EighJob), 256);
    lay.off_svd = o;
    o += make_layout(dtype, lay.svd_jobs.data(), n_jobs).total;
    lay.total = o;
    return lay;
}

}  // namespace

extern "C" int64_t tpa_eigh_worksize(int dtype, const int64_t *jobs_host, int n_jobs) {
    if (n_jobs <= 0) return 256;
    return make_eigh_layout(dtype, jobs_host, n_jobs).total;
}

extern "C" int tpa_eigh_batch(int dtype, const int64_t *jobs_host, int n_jobs, const void *a_base,
                              double *w_dev, void *v_base, void *work_dev, int64_t work_bytes,
                              int max_sweeps, double tol, int *sweeps_done, void *stream) {
    (void)tol;
    TPA_ARG_CHECK(dtype == TPA_F64 || dtype == TPA_C128);
    if (n_jobs <= 0) return 0;
    for (int b = 0; b < n_jobs; ++b) TPA_ARG_CHECK(jobs_host[8 * b + 1] > 0);
    EighLayout lay = make_eigh_layout(dtype, jobs_host, n_jobs);
    TPA_ARG_CHECK(work_bytes >= lay.total);
    hipStream_t st = (hipStream_t)stream;
    char *work = (char *)work_dev;
    EighJob *jobs = (EighJob *)(work + lay.off_jobs);
    double *mu = (double *)(work + lay.off_mu);
    TPA_HIP_CHECK(hipMemcpyAsync(jobs, lay.jobs.data(), lay.jobs.size() * sizeof(EighJob), hipMemcpyHostToDevice, st));
    Layout slay = make_layout(dtype, lay.svd_jobs.data(), n_jobs);
    int rc;
    if (dtype == TPA_F64) {
        eigh_shift_kernel<false><<<n_jobs, NT, 0, st>>>(jobs, (const double *)a_base, (double *)(work + lay.off_ap), mu);
        TPA_LAUNCH_CHECK();
        rc = svd_run<false>(slay, n_jobs, work + lay.off_ap, work + lay.off_u, (double *)(work + lay.off_s),
                            work + lay.off_vh, work + lay.off_svd, max_sweeps, sweeps_done, st);
        if (rc != 0) return rc;
        eigh_finish_kernel<false><<<n_jobs, NT, 0, st>>>(jobs, (const double *)(work + lay.off_u), (const double *)(work + lay.off_s), mu, w_dev, (double *)v_base);
    } else {
        eigh_shift_kernel<true><<<n_jobs, NT, 0, st>>>(jobs, (const double *)a_base, (double *)(work + lay.off_ap), mu);
        TPA_LAUNCH_CHECK();
        rc = svd_run<true>(slay, n_jobs, work + lay.off_ap, work + lay.off_u, (double *)(work + lay.off_s),
                           work + lay.off_vh, work + lay.off_svd, max_sweeps, sweeps_done, st);
        if (rc != 0) return rc;
        eigh_finish_kernel<true><<<n_jobs, NT, 0, st>>>(jobs, (const double *)(work + lay.off_u), (const double *)(work + lay.off_s), mu, w_dev, (double *)v_base);
    }
    TPA_LAUNCH_CHECK();
    TPA_HIP_CHECK(hipStreamSynchronize(st));
    return 0;
}
