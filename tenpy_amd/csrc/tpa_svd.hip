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
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <algorithm>
#include <cmath>
#include <memory>
#include <numeric>
#include <string>
#include <unordered_map>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

extern int tpa_qr_lookahead;      // tpa_qr.hip: real data, blocks of <= 2048 rows: one launch per panel (tpa_qr_la.inc)

namespace {

constexpr int NT = 256;  // 4 wavefronts = 4 row pairs per workgroup

struct SvdJob {  // int64[13], device copy
    int64_t w_off, g_off, R, L, Rpad, a_off, m, n, u_off, s_off, vh_off, sig_off;
    int64_t tr;   // 1: W = A^T (rows of W = columns of A), 0: W = A.  m == n: host flag bit 0 of jobs[6] selects W = A
    int64_t fro_bits;   // jobs[7]: IEEE-754 bits of a squared norm that replaces |A|_F^2 in the rank / floor decisions (0: none)
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
    const bool tr = (J.tr != 0);
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

// ||A||_F^2 per job (invariant under the rotations): the scale of the absolute part of the stopping rule.
// Two deterministic passes: FRO_PARTS workgroups per job, then one wavefront-sized sum.
constexpr int FRO_PARTS = 64;
template <bool CPLX>
__global__ __launch_bounds__(NT) void svd_fro_kernel(const SvdJob *__restrict__ jobs,
                                                     const double *__restrict__ A, double *__restrict__ fpart) {
    __shared__ double red[NT / 64];
    const SvdJob J = jobs[blockIdx.y];
    const int64_t tot = J.m * J.n * (CPLX ? 2 : 1);
    const double *a = A + (CPLX ? 2 : 1) * J.a_off;
    double s = 0;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < tot; e += (int64_t)FRO_PARTS * NT) s = fma(a[e], a[e], s);
    s = block_sum<NT>(s, red);
    if (threadIdx.x == 0) fpart[blockIdx.y * FRO_PARTS + blockIdx.x] = s;
}
__global__ __launch_bounds__(64) void svd_fro_sum_kernel(const SvdJob *__restrict__ jobs, const double *__restrict__ fpart,
                                                         double *__restrict__ fro2) {
    double s = fpart[blockIdx.x * FRO_PARTS + threadIdx.x];
    s = wave_sum(s);
    const int64_t ov = jobs[blockIdx.x].fro_bits;     // a NaN / Inf in the input still shows in s
    if (threadIdx.x == 0) fro2[blockIdx.x] = (ov != 0 && s == s && s < 1.0e300) ? __longlong_as_double(ov) : s;
}

// Stopping rule for a row pair with alpha=|x|^2 >= beta=|y|^2 (either order), gamma = x.conj(y):
//     |gamma| <= tol * sqrt(min(alpha,beta)) * max( sqrt(max(alpha,beta)), rho * ||A||_F )
// rho = 0 is the purely relative Hestenes/de Rijk criterion (cosine of the angle <= tol); rho > 0 adds
// an absolute floor: couplings that can change a singular value by less than ~tol*rho*||A||_F are left
// alone.  Without the floor the iteration keeps rotating rounding noise among rows whose norm is below
// ~eps*||A||_F/tol (strongly graded spectra, e.g. DMRG wave functions) and needs 5x more sweeps.
// Round 6: THE FLOOR ACTS ON THE SMALLER ROW (tpa_floor_on_min = 1, the default for callers that pass a negative rho, see
// tpa_svd_batch):  |gamma| <= tol * sqrt(max(alpha,beta)) * max( sqrt(min(alpha,beta)), rho * ||A||_F ),
// i.e. a pair stops at a cosine of tol * rho ||A|| / sigma_MIN instead of tol * rho ||A|| / sigma_MAX.  What that leaves behind is a
// component of the SMALLER row along the larger one of size <= tol rho ||A||: the absolute accuracy class again -- provided the
// post-processing that makes the normalised rows orthonormal moves only the smaller vector of a pair (ordered Gram-Schmidt clean-up,
// linalg/_svd_warm.py::ordered_rows) and not both (symmetric Loewdin, rounds 3 - 5: error sigma_max * cosine / 2, which is what forced
// the cosine down to tol rho ||A|| / sigma_max).  On the Jacobi inputs of the chi = 2048 sweeps two thirds of the block pairs never
// become active under this rule (profiles/r06_stopping_rule_emulation.txt).
// The mode travels in the SIGN of the floor: floor2 < 0 <=> floor |floor2| on the smaller row.  (A first version kept it in a __device__
// variable: one global load + s_waitcnt vmcnt(0) inside the angle chain of every elementary round.)
__device__ __forceinline__ double svd_floor2(double rho, double fro2) { return copysign(rho * rho * fro2, rho); }

__device__ __forceinline__ bool svd_needs_rotation(double a, double b, double g2, double tol, double floor2s) {
    if (!(a > 0.0) || !(b > 0.0)) return false;
    const bool tpa_floor_on_min = floor2s < 0.0;
    const double floor2 = fabs(floor2s);
    const double mn = fmin(a, b), mx0 = fmax(a, b);
    // A row 1e-30 times shorter than its partner (<= 1e-30 ||A||_F) is a zero row for every purpose.  Without this cut a
    // row that lies EXACTLY in the span of the others (exactly rank-deficient block with no room for rounding noise:
    // zero columns, bond matrices of the subspace expansion) shrinks by ~eps per sweep until |row|^2 is denormal;
    // there tol^2*mn*mx underflows to 0, zeta^2 overflows (rotation angle 0) and the pair is "rotated" forever.
    if (mn < 1.0e-60 * mx0) return false;
    if (tpa_floor_on_min) return g2 > tol * tol * mx0 * fmax(mn, floor2);
    const double mx = fmax(mx0, floor2);
    return g2 > tol * tol * mn * mx;
}

// The same test without control flow (the angle wavefront of the 32-row-block solve evaluates it inside its dependency chain).
__device__ __forceinline__ bool svd_needs_rotation_bf(double a, double b, double g2, double tol2, double floor2s) {
    const double mn = fmin(a, b), mx0 = fmax(a, b), floor2 = fabs(floor2s);
    const double thr = (floor2s < 0.0) ? mx0 * fmax(mn, floor2) : mn * fmax(mx0, floor2);
    return (mn > 0.0) & !(mn < 1.0e-60 * mx0) & (g2 > tol2 * thr);
}

// "Big" rotation: scaled cosine above 1e-7.  A sweep without any big rotation leaves all cosines at ~1e-14 or below
// (quadratic convergence), so the host may skip the verification sweep (tpa_svd_set_algorithm bit 10, off by default;
// tests/jacobi_emulation.py `predict`: one sweep of ~7 saved on chi=2048 theta blocks, identical singular values and
// orthogonality).  Counted in n_rot[1].
// Round 5: for a pair BELOW the floor the scale is sqrt(mx * floor^2), not floor^2.  The rotation leaves a cosine of ~cos^2, and that
// has to meet the pair's OWN stopping rule, |gamma'| <= tol sqrt(mn) rho |A|, i.e. cos^2 <~ tol rho |A| / sqrt(mx).  With floor^2 as the
// scale (rounds 2-4) a pair of rows with sigma ~ 1e-12 |A| was "big" only above cos = 1e-7 rho |A| / sigma >> 1: never -- the
// iteration could end on cosines of O(0.1) among such rows once the LARGE rows were done, which the first-order clean-up does not
// repair.  Harmless on pivoted-QR starts (those rows begin nearly orthogonal), visible on warm / sketch starts of blocks graded
// down to rounding level: tests/test_svd_configs_gpu.py found isometry defects of 1e-10 ... 3e-3 on the MI355X; the numpy emulation
// (tests/jacobi_emulation.py, NEW_BIG_RULE) shows 0.17 -> 1.8e-5 before the clean-up for rho = 1e-4 and 0.17 -> 1.2e-7 for rho = 1e-6.
__device__ __forceinline__ bool svd_big_rotation(double a, double b, double g2, double floor2s) {
    const double mn = fmin(a, b), mx0 = fmax(a, b);
    const bool tpa_floor_on_min = floor2s < 0.0;
    const double floor2 = fabs(floor2s);
    if (tpa_floor_on_min) {
        // the rotation leaves a cosine of ~cos^2, to be met by the pair's own rule: cos^2 <= tol rho |A| / sqrt(mn) below the floor
        if (mn >= floor2) return g2 > 1.0e-14 * mn * mx0;
        const double p = mn * floor2;
        return g2 > 1.0e-14 * mx0 * (p * __builtin_amdgcn_rsq(p));
    }
    if (mx0 >= floor2) return g2 > 1.0e-14 * mn * mx0;            // above the floor: the relative rule, as before
    const double p = mx0 * floor2;
    return g2 > 1.0e-14 * mn * (p * __builtin_amdgcn_rsq(p));     // sqrt(p) from the hardware seed: a threshold, not a result
}

template <bool CPLX>
__global__ __launch_bounds__(NT) void svd_round_kernel(const SvdJob *__restrict__ jobs,
                                                       const int2 *__restrict__ pairs, int round,
                                                       double *__restrict__ W, double *__restrict__ G,
                                                       unsigned int *__restrict__ n_rot,
                                                       const double *__restrict__ fro2, double rho) {
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
    if (!svd_needs_rotation(a, b, g2, tol, svd_floor2(rho, fro2[jp.x]))) return;  // already orthogonal (or NaN)
    const bool big_rot = svd_big_rotation(a, b, g2, svd_floor2(rho, fro2[jp.x]));
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
    if (lane == 0) {
        atomicAdd(n_rot, 1u);
        if (big_rot) atomicAdd(n_rot + 1, 1u);
    }
}

// ---------------------------------------------------------------------------------------------------
// Block one-sided Jacobi round (real): one workgroup owns a PAIR of row blocks (2 x 8 rows of W and of G).
//   phase 1  S = X X^T (16 x 16 Gram of the 16 rows) on the fp64 MFMA, K split over the 4 wavefronts,
//            row chunks staged through LDS with coalesced 512-B row segments;
//   phase 2  the 16 x 16 symmetric eigenproblem is solved completely inside the workgroup by cyclic
//            two-sided Jacobi (one matrix element per thread, round-robin pairs), accumulating the
//            row transform Q;  rotation angles are the Hestenes angles of the Gram entries;
//   phase 3  X <- Q X for the rows of W and of G, again as 16x16x4 MFMAs over LDS-staged chunks.
// Compared with one wavefront per row pair this needs 8x fewer rounds (launches) per sweep and converges
// in fewer sweeps (every 16-row subproblem is diagonalised exactly).  The Gram matrix is recomputed from
// the data in every round, so rounding errors of the in-LDS updates do not accumulate.
constexpr int BRJ = 8;            // rows per block
constexpr int TRJ = 2 * BRJ;      // rows per workgroup
constexpr int CHJ = 64;           // columns per staged chunk
constexpr int CHP = CHJ + 1;      // LDS row pitch of a chunk
struct BEntry {  // int32[4]
    int job, pair, part, nparts;
};

__device__ __forceinline__ void block_pair_of(const SvdJob &J, int pair, int round, int64_t &bi, int64_t &bj,
                                              int64_t &NB) {
    NB = (J.R + BRJ - 1) / BRJ;
    const int64_t NBp = (NB + 1) / 2 * 2;
    const int64_t mod = NBp - 1;
    const int64_t r = (mod > 0) ? (round % mod) : 0;
    if (pair == 0) {
        bi = NBp - 1;
        bj = r;
    } else {
        bi = (r + pair) % mod;
        bj = (r - pair + mod) % mod;
    }
    if (bi > bj) {
        const int64_t t = bi;
        bi = bj;
        bj = t;
    }
}

constexpr int NTG = 256;  // 4 wavefronts per (pair, part) workgroup

// Local solve of the 16 x 16 symmetric problem by ONE wavefront (callers: the first wavefront of the workgroup).
// Sm: Gram matrix (destroyed), Qm: accumulated rotations (identity on entry); csA / cpA / partA: LDS scratch.
__device__ __forceinline__ void svd_local_solve(double (*Sm)[TRJ + 1], double (*Qm)[TRJ + 1], double *csA, double *cpA,
                                                int *partA, int lane, int local_sweeps, int full_local, double tol,
                                                double floor2) {
    // ONE wavefront, 4 matrix elements per lane (row ei = lane / 4, columns ej0 .. ej0 + 3), no workgroup barriers.
    // `full` rounds sweep all 120 pairs of the 16 rows (15 local rounds); otherwise only the 64 CROSS pairs between the two
    // 8-row blocks are rotated (8 local rounds, partner of i < 8 is 8 + (i + rr) % 8): pairs inside a block were orthogonalised
    // when the block last took part in a full round and are only perturbed at second order since.  The host requests a full
    // round once per sweep for every pair.
    // A local round is a chain of LDS round trips, so it is kept short (measured in round 2: the solve was 7 of the 22 us of a
    // Jacobi round): every lane computes the rotation of ITS row itself (no hand-over of the row parameters), only the column
    // parameters go through LDS, and S <- R^T S R is applied in one pass,
    //     S'[i][j] = cs_i (c_j S[i][j] + s_j S[i][pj]) + cp_i (c_j S[pi][j] + s_j S[pi][pj]),
    // i.e. per local round: read 3 -> rotation -> write parameters -> read 36 -> write 8.
    const int ei = lane >> 2, ej0 = (lane & 3) * 4;
    const int n_local = full_local ? (TRJ - 1) : BRJ;
    for (int sweep = 0; sweep < local_sweeps; ++sweep) {
        bool rotated = false;
        for (int rr = 0; rr < n_local; ++rr) {
            int pi;
            if (full_local) {
                if (ei == TRJ - 1)
                    pi = rr;
                else if (ei == rr)
                    pi = TRJ - 1;
                else
                    pi = (2 * rr - ei + 2 * (TRJ - 1)) % (TRJ - 1);
            } else {
                pi = (ei < BRJ) ? (BRJ + ((ei + rr) & (BRJ - 1))) : (((ei - BRJ) - rr) & (BRJ - 1));
            }
            const int p = (ei < pi) ? ei : pi, q = (ei < pi) ? pi : ei;
            const double a = Sm[p][p], b = Sm[q][q], g = Sm[p][q];
            double cs = 1.0, cp = 0.0;
            bool rot_now = false;
            if (svd_needs_rotation(a, b, g * g, tol, floor2)) {
                const double zeta = (b - a) * __builtin_amdgcn_rcp(2.0 * g);
                const double h = __builtin_amdgcn_sqrt(fma(zeta, zeta, 1.0));
                const double t = copysign(1.0, zeta) * __builtin_amdgcn_rcp(fabs(zeta) + h);
                const double x = fma(t, t, 1.0);
                double c0 = __builtin_amdgcn_rsq(x);
                c0 = c0 * fma(-0.5 * x * c0, c0, 1.5);
                cs = c0 * fma(-0.5 * x * c0, c0, 1.5);
                const double sn = cs * t;
                cp = (ei == p) ? -sn : sn;
                rot_now = true;
            }
            if (!__any(rot_now)) continue;   // no pair of this local round needs a rotation
            rotated = true;
            if ((lane & 3) == 0) {
                partA[ei] = pi;
                csA[ei] = cs;
                cpA[ei] = cp;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            // all LDS reads of the pass are issued in two groups (the second depends on the partner indices of the first) and
            // only then consumed: left to itself the compiler waits for each of the 8 dependent reads separately
            int pjv[4];
            double cj[4], sj[4], s_ii[4], s_pi[4], q_i[4], q_p[4], s_ip[4], s_pp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                pjv[u] = partA[ej0 + u];
                cj[u] = csA[ej0 + u];
                sj[u] = cpA[ej0 + u];
                s_ii[u] = Sm[ei][ej0 + u];
                s_pi[u] = Sm[pi][ej0 + u];
                q_i[u] = Qm[ei][ej0 + u];
                q_p[u] = Qm[pi][ej0 + u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s_ip[u] = Sm[ei][pjv[u]];
                s_pp[u] = Sm[pi][pjv[u]];
            }
            __builtin_amdgcn_sched_barrier(0);
            double s_new[4], q_new[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double r0 = fma(cj[u], s_ii[u], sj[u] * s_ip[u]);
                const double r1 = fma(cj[u], s_pi[u], sj[u] * s_pp[u]);
                s_new[u] = fma(cs, r0, cp * r1);
                q_new[u] = fma(cs, q_i[u], cp * q_p[u]);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                Sm[ei][ej0 + u] = s_new[u];
                Qm[ei][ej0 + u] = q_new[u];
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
        if (!__any(rotated)) break;
    }
}


__global__ __launch_bounds__(NTG) void svd_gram_part_kernel(const SvdJob *__restrict__ jobs,
                                                            const BEntry *__restrict__ entries, int round,
                                                            const double *__restrict__ W,
                                                            double *__restrict__ gpart) {
    __shared__ double Xs[NTG / 64][TRJ][CHP];
    const BEntry E = entries[blockIdx.x];
    if (E.job < 0) return;
    const SvdJob J = jobs[E.job];
    int64_t bi, bj, NB;
    block_pair_of(J, E.pair, round, bi, bj, NB);
    const int64_t R = J.R, L = J.L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    int64_t rowoff[TRJ];
#pragma unroll
    for (int t = 0; t < TRJ; ++t) {
        const int64_t b = (t < BRJ) ? bi : bj;
        const int64_t r = b * BRJ + (t % BRJ);
        rowoff[t] = (b < NB && r < R) ? r : -1;
    }
    const int64_t nchunk = (L + CHJ - 1) / CHJ;
    const int64_t c_lo = nchunk * E.part / E.nparts, c_hi = nchunk * (E.part + 1) / E.nparts;
    d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    const double *Wb = W + J.w_off;
    double reg[TRJ];
    int64_t c = c_lo + wave;
    if (c < c_hi) {
        const int64_t col = c * CHJ + lane;
#pragma unroll
        for (int t = 0; t < TRJ; ++t) reg[t] = (rowoff[t] >= 0 && col < L) ? Wb[rowoff[t] * L + col] : 0.0;
    }
    for (; c < c_hi; c += NTG / 64) {
#pragma unroll
        for (int t = 0; t < TRJ; ++t) Xs[wave][t][lane] = reg[t];
        const int64_t cn = c + NTG / 64;
        if (cn < c_hi) {
            const int64_t col = cn * CHJ + lane;
#pragma unroll
            for (int t = 0; t < TRJ; ++t) reg[t] = (rowoff[t] >= 0 && col < L) ? Wb[rowoff[t] * L + col] : 0.0;
        }
#pragma unroll
        for (int ks = 0; ks < CHJ / 4; ks += 2) {
            const double a0 = Xs[wave][l15][ks * 4 + l4];
            const double a1 = Xs[wave][l15][ks * 4 + 4 + l4];
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, acc1, 0, 0, 0);
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) Xs[wave][l4 + 4 * r][l15] = acc0[r] + acc1[r];
    __syncthreads();
    {
        const int i = tid >> 4, j = tid & 15;
        double sacc = 0;
#pragma unroll
        for (int w = 0; w < NTG / 64; ++w) sacc += Xs[w][i][j];
        gpart[(int64_t)blockIdx.x * 256 + tid] = sacc;
    }
}

__global__ __launch_bounds__(NTG) void svd_solve_apply_kernel(const SvdJob *__restrict__ jobs,
                                                              const BEntry *__restrict__ entries, int round,
                                                              double *__restrict__ W, double *__restrict__ G,
                                                              const double *__restrict__ gpart,
                                                              unsigned int *__restrict__ n_rot,
                                                              const double *__restrict__ fro2, double rho,
                                                              int local_sweeps, int full_local) {
    __shared__ double Xs[NTG / 64][TRJ][CHP];
    __shared__ double Sm[TRJ][TRJ + 1], Qm[TRJ][TRJ + 1];
    __shared__ double csA[TRJ], cpA[TRJ];
    __shared__ int partA[TRJ];
    __shared__ int any_flag;
    const BEntry E = entries[blockIdx.x];
    if (E.job < 0) return;
    const SvdJob J = jobs[E.job];
    int64_t bi, bj, NB;
    block_pair_of(J, E.pair, round, bi, bj, NB);
    const int64_t R = J.R, L = J.L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    int64_t rowoff[TRJ];
#pragma unroll
    for (int t = 0; t < TRJ; ++t) {
        const int64_t b = (t < BRJ) ? bi : bj;
        const int64_t r = b * BRJ + (t % BRJ);
        rowoff[t] = (b < NB && r < R) ? r : -1;
    }
    // ---- full Gram = sum of the partials of this pair, fixed order (entries of a pair are consecutive)
    {
        const int64_t first = (int64_t)blockIdx.x - E.part;
        double sacc = 0;
        for (int p = 0; p < E.nparts; ++p) sacc += gpart[(first + p) * 256 + tid];
        Sm[tid >> 4][tid & 15] = sacc;
        Qm[tid >> 4][tid & 15] = ((tid >> 4) == (tid & 15)) ? 1.0 : 0.0;
    }
    if (tid == 0) any_flag = 0;
    __syncthreads();
    const double tol = 2.220446049250313e-16 * sqrt((double)L);
    const double floor2 = svd_floor2(rho, fro2[E.job]);
    {
        const int ei = tid >> 4, ej = tid & 15;
        const bool relevant = full_local ? (ei < ej) : (ei < BRJ && ej >= BRJ);
        if (relevant && svd_needs_rotation(Sm[ei][ei], Sm[ej][ej], Sm[ei][ej] * Sm[ei][ej], tol, floor2))
            atomicOr(&any_flag, svd_big_rotation(Sm[ei][ei], Sm[ej][ej], Sm[ei][ej] * Sm[ei][ej], floor2) ? 3 : 1);
    }
    __syncthreads();
    if (any_flag == 0) return;
    if (tid == 0 && E.part == 0) {
        atomicAdd(n_rot, 1u);
        if (any_flag & 2) atomicAdd(n_rot + 1, 1u);
    }

    if (wave == 0) svd_local_solve(Sm, Qm, csA, cpA, partA, lane, local_sweeps, full_local, tol, floor2);
    __syncthreads();

    // ---- apply: X <- Q X on this part's columns of the W rows and of the G rows.  The column chunks of W and of G
    //      assigned to this part form ONE list that is dealt out to the wavefronts, so that with enough parts every
    //      wavefront has a single chunk: one load round trip, 16 MFMAs, one store (no second, dependent pass for G).
    double qa[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qa[kk] = Qm[l15][kk * 4 + l4];
    const int64_t nchW = (L + CHJ - 1) / CHJ, nchG = (R + CHJ - 1) / CHJ;
    const int64_t w_lo = nchW * E.part / E.nparts, w_hi = nchW * (E.part + 1) / E.nparts;
    const int64_t g_lo = nchG * E.part / E.nparts, g_hi = nchG * (E.part + 1) / E.nparts;
    const int64_t nW = w_hi - w_lo, nU = nW + (g_hi - g_lo);
    for (int64_t u = wave; u < nU; u += NTG / 64) {
        const bool isW = u < nW;
        double *M = isW ? (W + J.w_off) : (G + J.g_off);
        const int64_t len = isW ? L : R;
        const int64_t c = isW ? (w_lo + u) : (g_lo + (u - nW));
        const int64_t col = c * CHJ + lane;
        double reg[TRJ];
#pragma unroll
        for (int t = 0; t < TRJ; ++t) reg[t] = (rowoff[t] >= 0 && col < len) ? M[rowoff[t] * len + col] : 0.0;
#pragma unroll
        for (int t = 0; t < TRJ; ++t) Xs[wave][t][lane] = reg[t];
        d4 o[CHJ / 16];
#pragma unroll
        for (int tile = 0; tile < CHJ / 16; ++tile) o[tile] = d4{0, 0, 0, 0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int tile = 0; tile < CHJ / 16; ++tile) {
                const double bb = Xs[wave][kk * 4 + l4][tile * 16 + l15];
                o[tile] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[kk], bb, o[tile], 0, 0, 0);
            }
#pragma unroll
        for (int tile = 0; tile < CHJ / 16; ++tile) {
            const int64_t oc = c * CHJ + tile * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gr = rowoff[l4 + 4 * r];
                if (gr >= 0 && oc < len) M[gr * len + oc] = o[tile][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// One launch per round: Gram, local solve and application fused.  Each (pair, column part) workgroup loads its
// column chunks of the 16 W rows (and of the 16 G rows) ONCE into LDS, publishes the partial Gram of its W columns,
// waits for the partials of the other parts of the same pair (a counter per pair: agent-scope release / acquire,
// only the <= 8 sibling workgroups of a pair synchronise, never the grid), solves the 16 x 16 problem and applies
// Q to the LDS-resident chunks.  Compared with the two-kernel round this removes one kernel boundary, one table /
// row reload and one pass over W.  Requirements (checked by the host, else the two-kernel round is used): every
// workgroup of the launch is resident at the same time (spin wait!) and a part has at most 4 * FIT chunks.
// A bounded spin (FUSED_SPIN_MAX polls) turns a lost sibling into an error code instead of a hang.
// Up to 8 agent-coherent (sc1) loads issued back to back, ONE wait: reads data that sibling workgroups published with
// write-through stores.  (A chain of __hip_atomic_load costs one memory round trip EACH; an acquire fence + plain loads
// invalidates the XCD's L2 under every other workgroup: +25 % on the whole SVD.)
__device__ __forceinline__ double sum_coherent(const double *base, int64_t stride, int n) {
    double v[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        v[p] = 0.0;
        if (p < n) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v[p]) : "v"(base + p * stride) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                 :
                 : "memory");
    double t = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) t += v[p];      // fixed order; the slots p >= n hold 0
    return t;
}

constexpr int FIT = 2;                      // chunks per wavefront kept in LDS
constexpr unsigned FUSED_SPIN_MAX = 4000000u;

__global__ __launch_bounds__(NTG, 4) void svd_round_fused_kernel(const SvdJob *__restrict__ jobs,
                                                                 const BEntry *__restrict__ entries, int round,
                                                                 double *__restrict__ W, double *__restrict__ G,
                                                                 double *gpart, unsigned int *pair_cnt, unsigned int seq,
                                                                 unsigned int *__restrict__ n_rot,
                                                                 const double *__restrict__ fro2, double rho,
                                                                 int local_sweeps, int full_local, int *err_flag) {
    // Occupancy is the point of this layout: the chunks of a part live in REGISTERS between the Gram and the application
    // (FIT x 16 doubles per lane) and pass through ONE per-wavefront LDS staging tile for the MFMA operand layouts, so a
    // workgroup needs 38 KB of LDS and <= 128 VGPRs -> 4 workgroups per CU, 1024 on the chip.  A chi = 2048 theta has ~970
    // (pair, part) entries per round: with the previous 80 KB layout (2 per CU) every round ran as two waves of workgroups
    // (measured in round 2: 17.6 of the 25 us of a round were there with Gram exchange and local solve removed).
    __shared__ double Xs[NTG / 64][TRJ][CHP];
    __shared__ double Sm[TRJ][TRJ + 1], Qm[TRJ][TRJ + 1];
    __shared__ double csA[TRJ], cpA[TRJ];
    __shared__ int partA[TRJ];
    __shared__ int any_flag;
    const BEntry E = entries[blockIdx.x];
    if (E.job < 0) return;
    const SvdJob J = jobs[E.job];
    int64_t bi, bj, NB;
    block_pair_of(J, E.pair, round, bi, bj, NB);
    const int R = (int)J.R, L = (int)J.L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int row_i = (bi < NB) ? (int)bi * BRJ : R, row_j = (bj < NB) ? (int)bj * BRJ : R;      // row t: (t < 8 ? row_i : row_j) + t % 8
    const int nchW = (L + CHJ - 1) / CHJ, nchG = (R + CHJ - 1) / CHJ;
    const int w_lo = (int)((int64_t)nchW * E.part / E.nparts), w_hi = (int)((int64_t)nchW * (E.part + 1) / E.nparts);
    const int g_lo = (int)((int64_t)nchG * E.part / E.nparts), g_hi = (int)((int64_t)nchG * (E.part + 1) / E.nparts);
    const int nW = w_hi - w_lo, nU = nW + (g_hi - g_lo);
    double (*Xw)[CHP] = Xs[wave];
    // ---- load all chunks of this part (both iterations in flight)
    double reg[FIT][TRJ];
#pragma unroll
    for (int it = 0; it < FIT; ++it) {
        const int u = wave + it * (NTG / 64);
        const bool isW = u < nW;
        const double *M = isW ? (W + J.w_off) : (G + J.g_off);
        const int len = isW ? L : R;
        const int col = (isW ? (w_lo + u) : (g_lo + (u - nW))) * CHJ + lane;
        const bool ok = u < nU && col < len;
#pragma unroll
        for (int t = 0; t < TRJ; ++t) {
            const int row = ((t < BRJ) ? row_i : row_j) + (t % BRJ);
            reg[it][t] = (ok && row < R) ? M[(int64_t)row * len + col] : 0.0;
        }
    }
    // ---- partial Gram of the W chunks
    d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
    for (int it = 0; it < FIT; ++it) {
        const int u = wave + it * (NTG / 64);
        if (u < nW) {   // wave-uniform
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < TRJ; ++t) Xw[t][lane] = reg[it][t];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ks = 0; ks < CHJ / 4; ks += 2) {
                const double a0 = Xw[l15][ks * 4 + l4];
                const double a1 = Xw[l15][ks * 4 + 4 + l4];
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, acc1, 0, 0, 0);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // the staging tile of this wavefront now carries its 16 x 16 partial (pitch TRJ + 1) to the cross-wave sum
    double (*Gw)[TRJ + 1] = reinterpret_cast<double (*)[TRJ + 1]>(&Xs[wave][0][0]);
#pragma unroll
    for (int r = 0; r < 4; ++r) Gw[l4 + 4 * r][l15] = acc0[r] + acc1[r];
    if (tid == 0) any_flag = 0;
    __syncthreads();
    const int64_t first = (int64_t)blockIdx.x - E.part;
    {
        const int i = tid >> 4, j = tid & 15;
        double sacc = 0;
#pragma unroll
        for (int w = 0; w < NTG / 64; ++w) sacc += reinterpret_cast<double (*)[TRJ + 1]>(&Xs[w][0][0])[i][j];
        // write-through store (agent scope -> sc1): the partial is not left dirty in this XCD's L2, so no cache
        // write-back fence is needed before the flag (a __threadfence() here cost ~45 us per round with ~400
        // workgroups fencing at once)
        __hip_atomic_store(&gpart[(int64_t)blockIdx.x * 256 + tid], sacc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- publish the partial, wait for the siblings of this pair
    if (E.nparts > 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my store has reached the coherent level
        __syncthreads();
        if (tid == 0) {
            atomicAdd(&pair_cnt[first], 1u);
            const unsigned int target = (unsigned int)E.nparts * seq;
            unsigned int spins = 0;
            while (__hip_atomic_load(&pair_cnt[first], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > FUSED_SPIN_MAX) {
                    atomicExch(err_flag, 1);
                    break;
                }
            }
        }
        __syncthreads();
    }
    // ---- full Gram = sum of the partials in a fixed order (identical in every part)
    {
        const double sacc = sum_coherent(gpart + first * 256 + tid, 256, E.nparts);   // never served from a stale L1 / foreign-L2 line
        Sm[tid >> 4][tid & 15] = sacc;
        Qm[tid >> 4][tid & 15] = ((tid >> 4) == (tid & 15)) ? 1.0 : 0.0;
    }
    __syncthreads();
    const double tol = 2.220446049250313e-16 * sqrt((double)L);
    const double floor2 = svd_floor2(rho, fro2[E.job]);
    {
        const int ei = tid >> 4, ej = tid & 15;
        const bool relevant = full_local ? (ei < ej) : (ei < BRJ && ej >= BRJ);
        if (relevant && svd_needs_rotation(Sm[ei][ei], Sm[ej][ej], Sm[ei][ej] * Sm[ei][ej], tol, floor2))
            atomicOr(&any_flag, svd_big_rotation(Sm[ei][ei], Sm[ej][ej], Sm[ei][ej] * Sm[ei][ej], floor2) ? 3 : 1);
    }
    __syncthreads();
    if (any_flag == 0) return;
    if (tid == 0 && E.part == 0) {
        atomicAdd(n_rot, 1u);
        if (any_flag & 2) atomicAdd(n_rot + 1, 1u);
    }
    if (wave == 0) svd_local_solve(Sm, Qm, csA, cpA, partA, lane, local_sweeps, full_local, tol, floor2);
    __syncthreads();
    // ---- apply Q to the register-resident chunks (through the staging tile again, now in the B-operand layout)
    double qa[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qa[kk] = Qm[l15][kk * 4 + l4];
#pragma unroll
    for (int it = 0; it < FIT; ++it) {
        const int u = wave + it * (NTG / 64);
        if (u >= nU) continue;
        const bool isW = u < nW;
        double *M = isW ? (W + J.w_off) : (G + J.g_off);
        const int len = isW ? L : R;
        const int c0 = (isW ? (w_lo + u) : (g_lo + (u - nW))) * CHJ;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < TRJ; ++t) Xw[t][lane] = reg[it][t];
        __builtin_amdgcn_wave_barrier();
        d4 o[CHJ / 16];
#pragma unroll
        for (int tile = 0; tile < CHJ / 16; ++tile) o[tile] = d4{0, 0, 0, 0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int tile = 0; tile < CHJ / 16; ++tile) {
                const double bb = Xw[kk * 4 + l4][tile * 16 + l15];
                o[tile] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[kk], bb, o[tile], 0, 0, 0);
            }
#pragma unroll
        for (int tile = 0; tile < CHJ / 16; ++tile) {
            const int oc = c0 + tile * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = l4 + 4 * r;
                const int row = ((t < BRJ) ? row_i : row_j) + (t % BRJ);
                if (row < R && oc < len) M[(int64_t)row * len + oc] = o[tile][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// One WORKGROUP per pair: 16 wavefronts x FIT chunks cover up to 32 chunks = 2048 columns of [W | G], which is every
// block the rank-revealing QR leaves of a chi <= 2048 DMRG theta (r + N <= 570 + 1084).  The partial Grams of the
// wavefronts meet in LDS, so the whole exchange of the fused round (write-through store, counter, spin, coherent
// reload: ~4 of its ~22 us) and the redundant local solves of the sibling parts disappear, and no workgroup ever
// waits for another one (no co-residency requirement).  Same arithmetic in the same order as the fused round with
// nparts = 1 except for the grouping of the partial sums (16 wavefronts instead of parts x 4).
constexpr int NTW = 1024;

__global__ __launch_bounds__(NTW) void svd_round_wide_kernel(const SvdJob *__restrict__ jobs, const int2 *__restrict__ wpairs,
                                                             int round, double *__restrict__ W, double *__restrict__ G,
                                                             unsigned int *__restrict__ n_rot, const double *__restrict__ fro2,
                                                             double rho, int local_sweeps, int full_local) {
    __shared__ double Xs[NTW / 64][TRJ][CHP];
    __shared__ double Sm[TRJ][TRJ + 1], Qm[TRJ][TRJ + 1];
    __shared__ double csA[TRJ], cpA[TRJ];
    __shared__ int partA[TRJ];
    __shared__ int any_flag;
    const int2 E = wpairs[blockIdx.x];
    if (E.x < 0) return;
    const SvdJob J = jobs[E.x];
    int64_t bi, bj, NB;
    block_pair_of(J, E.y, round, bi, bj, NB);
    const int R = (int)J.R, L = (int)J.L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int row_i = (bi < NB) ? (int)bi * BRJ : R, row_j = (bj < NB) ? (int)bj * BRJ : R;
    const int nW = (L + CHJ - 1) / CHJ, nU = nW + (R + CHJ - 1) / CHJ;       // host guarantees nU <= FIT * NTW / 64
    double (*Xw)[CHP] = Xs[wave];
    double reg[FIT][TRJ];
#pragma unroll
    for (int it = 0; it < FIT; ++it) {
        const int u = wave + it * (NTW / 64);
        const bool isW = u < nW;
        const double *M = isW ? (W + J.w_off) : (G + J.g_off);
        const int len = isW ? L : R;
        const int col = (isW ? u : (u - nW)) * CHJ + lane;
        const bool ok = u < nU && col < len;
#pragma unroll
        for (int t = 0; t < TRJ; ++t) {
            const int row = ((t < BRJ) ? row_i : row_j) + (t % BRJ);
            reg[it][t] = (ok && row < R) ? M[(int64_t)row * len + col] : 0.0;
        }
    }
    d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
    for (int it = 0; it < FIT; ++it) {
        const int u = wave + it * (NTW / 64);
        if (u < nW) {   // wave-uniform
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < TRJ; ++t) Xw[t][lane] = reg[it][t];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ks = 0; ks < CHJ / 4; ks += 2) {
                const double a0 = Xw[l15][ks * 4 + l4];
                const double a1 = Xw[l15][ks * 4 + 4 + l4];
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, acc1, 0, 0, 0);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    double (*Gw)[TRJ + 1] = reinterpret_cast<double (*)[TRJ + 1]>(&Xs[wave][0][0]);
#pragma unroll
    for (int r = 0; r < 4; ++r) Gw[l4 + 4 * r][l15] = acc0[r] + acc1[r];
    if (tid == 0) any_flag = 0;
    __syncthreads();
    const double tol = 2.220446049250313e-16 * sqrt((double)L);
    const double floor2 = svd_floor2(rho, fro2[E.x]);
    if (tid < TRJ * TRJ) {
        const int i = tid >> 4, j = tid & 15;
        double sacc = 0;
#pragma unroll
        for (int w = 0; w < NTW / 64; ++w) sacc += reinterpret_cast<double (*)[TRJ + 1]>(&Xs[w][0][0])[i][j];   // fixed order
        Sm[i][j] = sacc;
        Qm[i][j] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();
    if (tid < TRJ * TRJ) {
        const int ei = tid >> 4, ej = tid & 15;
        const bool relevant = full_local ? (ei < ej) : (ei < BRJ && ej >= BRJ);
        if (relevant && svd_needs_rotation(Sm[ei][ei], Sm[ej][ej], Sm[ei][ej] * Sm[ei][ej], tol, floor2))
            atomicOr(&any_flag, svd_big_rotation(Sm[ei][ei], Sm[ej][ej], Sm[ei][ej] * Sm[ei][ej], floor2) ? 3 : 1);
    }
    __syncthreads();
    if (any_flag == 0) return;
    if (tid == 0) {
        atomicAdd(n_rot, 1u);
        if (any_flag & 2) atomicAdd(n_rot + 1, 1u);
    }
    if (wave == 0) svd_local_solve(Sm, Qm, csA, cpA, partA, lane, local_sweeps, full_local, tol, floor2);
    __syncthreads();
    double qa[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qa[kk] = Qm[l15][kk * 4 + l4];
#pragma unroll
    for (int it = 0; it < FIT; ++it) {
        const int u = wave + it * (NTW / 64);
        if (u >= nU) continue;
        const bool isW = u < nW;
        double *M = isW ? (W + J.w_off) : (G + J.g_off);
        const int len = isW ? L : R;
        const int c0 = (isW ? u : (u - nW)) * CHJ;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < TRJ; ++t) Xw[t][lane] = reg[it][t];
        __builtin_amdgcn_wave_barrier();
        d4 o[CHJ / 16];
#pragma unroll
        for (int tile = 0; tile < CHJ / 16; ++tile) o[tile] = d4{0, 0, 0, 0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int tile = 0; tile < CHJ / 16; ++tile) {
                const double bb = Xw[kk * 4 + l4][tile * 16 + l15];
                o[tile] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[kk], bb, o[tile], 0, 0, 0);
            }
#pragma unroll
        for (int tile = 0; tile < CHJ / 16; ++tile) {
            const int oc = c0 + tile * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = l4 + 4 * r;
                const int row = ((t < BRJ) ? row_i : row_j) + (t % BRJ);
                if (row < R && oc < len) M[(int64_t)row * len + oc] = o[tile][r];
            }
        }
    }
}

#include "tpa_svd_b32.inc"
#include "tpa_svd_b32c.inc"
#include "tpa_svd_gram.inc"

// ---------------------------------------------------------------------------------------------------
// Complex (Hermitian) version of the split block-Jacobi round.  Same structure; chunks are 32 complex columns
// (512 B per row segment), re / im planes in LDS, 4 real MFMAs per complex product.
//   S = X X^H :  Re S = Xr Xr^T + Xi Xi^T ,  Im S = Xi Xr^T - Xr Xi^T
//   rotation of rows (p, q) with gamma = S_pq = |gamma| e^{i phi}:
//       x_p' = c x_p - s e^{i phi} x_q ,   x_q' = s e^{-i phi} x_p + c x_q          (same as svd_round_kernel<true>)
constexpr int CHC = 32;
constexpr int CHCP = CHC + 1;

__global__ __launch_bounds__(NTG) void svd_gram_part_kernel_c(const SvdJob *__restrict__ jobs,
                                                              const BEntry *__restrict__ entries, int round,
                                                              const double2 *__restrict__ W,
                                                              double *__restrict__ gpart) {
    __shared__ double Xr[NTG / 64][TRJ][CHCP], Xi[NTG / 64][TRJ][CHCP];
    const BEntry E = entries[blockIdx.x];
    if (E.job < 0) return;
    const SvdJob J = jobs[E.job];
    int64_t bi, bj, NB;
    block_pair_of(J, E.pair, round, bi, bj, NB);
    const int64_t R = J.R, L = J.L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int half = lane >> 5, l31 = lane & 31;  // two rows per load instruction: lanes 0-31 row t, 32-63 row t+1
    int64_t rowoff[TRJ];
#pragma unroll
    for (int t = 0; t < TRJ; ++t) {
        const int64_t b = (t < BRJ) ? bi : bj;
        const int64_t r = b * BRJ + (t % BRJ);
        rowoff[t] = (b < NB && r < R) ? r : -1;
    }
    const int64_t nchunk = (L + CHC - 1) / CHC;
    const int64_t c_lo = nchunk * E.part / E.nparts, c_hi = nchunk * (E.part + 1) / E.nparts;
    d4 ar0 = {0, 0, 0, 0}, ai0 = {0, 0, 0, 0};
    const double2 *Wb = W + J.w_off;
    for (int64_t c = c_lo + wave; c < c_hi; c += NTG / 64) {
        const int64_t col = c * CHC + l31;
#pragma unroll
        for (int t = 0; t < TRJ; t += 2) {
            const int64_t ro = half ? rowoff[t + 1] : rowoff[t];
            const double2 v = (ro >= 0 && col < L) ? Wb[ro * L + col] : double2{0, 0};
            Xr[wave][t + half][l31] = v.x;
            Xi[wave][t + half][l31] = v.y;
        }
#pragma unroll
        for (int ks = 0; ks < CHC / 4; ++ks) {
            const double xr = Xr[wave][l15][ks * 4 + l4], xi = Xi[wave][l15][ks * 4 + l4];
            ar0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xr, xr, ar0, 0, 0, 0);
            ar0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xi, xi, ar0, 0, 0, 0);
            ai0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xi, xr, ai0, 0, 0, 0);
            ai0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-xr, xi, ai0, 0, 0, 0);
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        Xr[wave][l4 + 4 * r][l15] = ar0[r];
        Xi[wave][l4 + 4 * r][l15] = ai0[r];
    }
    __syncthreads();
    {
        const int i = tid >> 4, j = tid & 15;
        double sr = 0, si = 0;
#pragma unroll
        for (int w = 0; w < NTG / 64; ++w) {
            sr += Xr[w][i][j];
            si += Xi[w][i][j];
        }
        gpart[(int64_t)blockIdx.x * 512 + tid] = sr;
        gpart[(int64_t)blockIdx.x * 512 + 256 + tid] = si;
    }
}

// Local solve of the 16 x 16 Hermitian problem by ONE wavefront (complex counterpart of svd_local_solve).
__device__ __forceinline__ void svd_local_solve_c(double (*Sr)[TRJ + 1], double (*Si)[TRJ + 1], double (*Qr)[TRJ + 1],
                                                  double (*Qi)[TRJ + 1], double *csA, double *cprA, double *cpiA, int *partA,
                                                  int lane, int local_sweeps, int full_local, double tol, double floor2) {
        const int ei = lane >> 2, ej0 = (lane & 3) * 4;
        const int n_local = full_local ? (TRJ - 1) : BRJ;
        for (int sweep = 0; sweep < local_sweeps; ++sweep) {
            bool rotated = false;
            for (int rr = 0; rr < n_local; ++rr) {
                if (lane < TRJ) {
                    const int i = lane;
                    int pi;
                    if (full_local) {
                        if (i == TRJ - 1)
                            pi = rr;
                        else if (i == rr)
                            pi = TRJ - 1;
                        else
                            pi = (2 * rr - i + 2 * (TRJ - 1)) % (TRJ - 1);
                    } else {
                        pi = (i < BRJ) ? (BRJ + ((i + rr) & (BRJ - 1))) : (((i - BRJ) - rr) & (BRJ - 1));
                    }
                    const int p = (i < pi) ? i : pi, q = (i < pi) ? pi : i;
                    const double a = Sr[p][p], b = Sr[q][q], gr = Sr[p][q], gi = Si[p][q];
                    const double g2 = gr * gr + gi * gi;
                    double c = 1.0, sre = 0.0, sim = 0.0;   // s e^{i phi}
                    if (svd_needs_rotation(a, b, g2, tol, floor2)) {
                        const double gabs = sqrt(g2);
                        const double ig = __builtin_amdgcn_rcp(gabs);
                        const double zeta = (b - a) * 0.5 * ig;
                        const double h = __builtin_amdgcn_sqrt(fma(zeta, zeta, 1.0));
                        const double t = copysign(1.0, zeta) * __builtin_amdgcn_rcp(fabs(zeta) + h);
                        const double x = fma(t, t, 1.0);
                        double c0 = __builtin_amdgcn_rsq(x);
                        c0 = c0 * fma(-0.5 * x * c0, c0, 1.5);
                        c = c0 * fma(-0.5 * x * c0, c0, 1.5);
                        const double s = c * t;
                        // unit phase to full precision: (gr, gi) / |gamma| with a Newton-refined reciprocal
                        double ig2 = ig * (2.0 - gabs * ig);
                        ig2 = ig2 * (2.0 - gabs * ig2);
                        sre = s * gr * ig2;
                        sim = s * gi * ig2;
                        rotated = true;
                    }
                    partA[i] = pi;
                    csA[i] = c;
                    // row p: x' = c x - s e^{i phi} y ; row q: y' = s e^{-i phi} x + c y
                    cprA[i] = (i == p) ? -sre : sre;
                    cpiA[i] = (i == p) ? -sim : -sim;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                {
                    const int pi = partA[ei];
                    const double cs = csA[ei], cr = cprA[ei], ci = cpiA[ei];
                    double nsr[4], nsi[4], nqr[4], nqi[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int ej = ej0 + u;
                        const double yr = Sr[pi][ej], yi = Si[pi][ej];
                        nsr[u] = cs * Sr[ei][ej] + (cr * yr - ci * yi);
                        nsi[u] = cs * Si[ei][ej] + (cr * yi + ci * yr);
                        const double zr = Qr[pi][ej], zi = Qi[pi][ej];
                        nqr[u] = cs * Qr[ei][ej] + (cr * zr - ci * zi);
                        nqi[u] = cs * Qi[ei][ej] + (cr * zi + ci * zr);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        Sr[ei][ej0 + u] = nsr[u];
                        Si[ei][ej0 + u] = nsi[u];
                        Qr[ei][ej0 + u] = nqr[u];
                        Qi[ei][ej0 + u] = nqi[u];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                {
                    // columns: S <- S J^H : S[.][j] = S[.][j] conj(cs_j) + S[.][pj] conj(cp_j)
                    double nsr[4], nsi[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int ej = ej0 + u, pj = partA[ej];
                        const double cr = cprA[ej], ci = -cpiA[ej];
                        const double yr = Sr[ei][pj], yi = Si[ei][pj];
                        nsr[u] = csA[ej] * Sr[ei][ej] + (cr * yr - ci * yi);
                        nsi[u] = csA[ej] * Si[ei][ej] + (cr * yi + ci * yr);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        Sr[ei][ej0 + u] = nsr[u];
                        Si[ei][ej0 + u] = nsi[u];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            }
            if (!__any(rotated)) break;
        }
}

__global__ __launch_bounds__(NTG) void svd_solve_apply_kernel_c(const SvdJob *__restrict__ jobs,
                                                                const BEntry *__restrict__ entries, int round,
                                                                double2 *__restrict__ W, double2 *__restrict__ G,
                                                                const double *__restrict__ gpart,
                                                                unsigned int *__restrict__ n_rot,
                                                                const double *__restrict__ fro2, double rho,
                                                                int local_sweeps, int full_local) {
    __shared__ double Xr[NTG / 64][TRJ][CHCP], Xi[NTG / 64][TRJ][CHCP];
    __shared__ double Sr[TRJ][TRJ + 1], Si[TRJ][TRJ + 1], Qr[TRJ][TRJ + 1], Qi[TRJ][TRJ + 1];
    __shared__ double csA[TRJ], cprA[TRJ], cpiA[TRJ];
    __shared__ int partA[TRJ];
    __shared__ int any_flag;
    const BEntry E = entries[blockIdx.x];
    if (E.job < 0) return;
    const SvdJob J = jobs[E.job];
    int64_t bi, bj, NB;
    block_pair_of(J, E.pair, round, bi, bj, NB);
    const int64_t R = J.R, L = J.L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int half = lane >> 5, l31 = lane & 31;
    int64_t rowoff[TRJ];
#pragma unroll
    for (int t = 0; t < TRJ; ++t) {
        const int64_t b = (t < BRJ) ? bi : bj;
        const int64_t r = b * BRJ + (t % BRJ);
        rowoff[t] = (b < NB && r < R) ? r : -1;
    }
    {
        const int64_t first = (int64_t)blockIdx.x - E.part;
        double sr = 0, si = 0;
        for (int p = 0; p < E.nparts; ++p) {
            sr += gpart[(first + p) * 512 + tid];
            si += gpart[(first + p) * 512 + 256 + tid];
        }
        Sr[tid >> 4][tid & 15] = sr;
        Si[tid >> 4][tid & 15] = si;
        Qr[tid >> 4][tid & 15] = ((tid >> 4) == (tid & 15)) ? 1.0 : 0.0;
        Qi[tid >> 4][tid & 15] = 0.0;
    }
    if (tid == 0) any_flag = 0;
    __syncthreads();
    const double tol = 2.220446049250313e-16 * sqrt((double)L);
    const double floor2 = svd_floor2(rho, fro2[E.job]);
    {
        const int ei = tid >> 4, ej = tid & 15;
        const bool relevant = full_local ? (ei < ej) : (ei < BRJ && ej >= BRJ);
        if (relevant && svd_needs_rotation(Sr[ei][ei], Sr[ej][ej], Sr[ei][ej] * Sr[ei][ej] + Si[ei][ej] * Si[ei][ej], tol, floor2))
            atomicOr(&any_flag, svd_big_rotation(Sr[ei][ei], Sr[ej][ej], Sr[ei][ej] * Sr[ei][ej] + Si[ei][ej] * Si[ei][ej], floor2) ? 3 : 1);
    }
    __syncthreads();
    if (any_flag == 0) return;
    if (tid == 0 && E.part == 0) {
        atomicAdd(n_rot, 1u);
        if (any_flag & 2) atomicAdd(n_rot + 1, 1u);
    }

    if (wave == 0) svd_local_solve_c(Sr, Si, Qr, Qi, csA, cprA, cpiA, partA, lane, local_sweeps, full_local, tol, floor2);
    __syncthreads();

    // ---- apply X <- Q X (complex): Xr' = Qr Xr - Qi Xi ; Xi' = Qr Xi + Qi Xr
    double qar[4], qai[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        qar[kk] = Qr[l15][kk * 4 + l4];
        qai[kk] = Qi[l15][kk * 4 + l4];
    }
    for (int pass = 0; pass < 2; ++pass) {
        double2 *M = (pass == 0) ? (W + J.w_off) : (G + J.g_off);
        const int64_t len = (pass == 0) ? L : R;
        const int64_t nch = (len + CHC - 1) / CHC;
        const int64_t c_lo = nch * E.part / E.nparts, c_hi = nch * (E.part + 1) / E.nparts;
        for (int64_t c = c_lo + wave; c < c_hi; c += NTG / 64) {
            const int64_t col = c * CHC + l31;
#pragma unroll
            for (int t = 0; t < TRJ; t += 2) {
                const int64_t ro = half ? rowoff[t + 1] : rowoff[t];
                const double2 v = (ro >= 0 && col < len) ? M[ro * len + col] : double2{0, 0};
                Xr[wave][t + half][l31] = v.x;
                Xi[wave][t + half][l31] = v.y;
            }
#pragma unroll
            for (int tile = 0; tile < CHC / 16; ++tile) {
                d4 orr = {0, 0, 0, 0}, oii = {0, 0, 0, 0};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const double br = Xr[wave][kk * 4 + l4][tile * 16 + l15], bim = Xi[wave][kk * 4 + l4][tile * 16 + l15];
                    orr = __builtin_amdgcn_mfma_f64_16x16x4f64(qar[kk], br, orr, 0, 0, 0);
                    orr = __builtin_amdgcn_mfma_f64_16x16x4f64(-qai[kk], bim, orr, 0, 0, 0);
                    oii = __builtin_amdgcn_mfma_f64_16x16x4f64(qar[kk], bim, oii, 0, 0, 0);
                    oii = __builtin_amdgcn_mfma_f64_16x16x4f64(qai[kk], br, oii, 0, 0, 0);
                }
                const int64_t oc = c * CHC + tile * 16 + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t gr = rowoff[l4 + 4 * r];
                    if (gr >= 0 && oc < len) M[gr * len + oc] = double2{orr[r], oii[r]};
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Complex fused round: Gram, sibling exchange, local solve and application in ONE launch -- the complex counterpart of
// svd_round_fused_kernel (same hand-off protocol: write-through partials, one counter per pair, bounded spin).  A part's chunks
// (32 complex columns each, <= 2 per wavefront) stay in registers between Gram and application; the 16 x 16 matrices S and Q
// live in the staging tiles of wavefronts 2 and 3 while those are idle, so a workgroup needs 34 KB of LDS (4 per CU).
// TEBD at chi = 1024 (two 1024 x 1024 complex blocks): the two-kernel round cost 15.5 + 38.7 us.
constexpr int FITC = 2;

__device__ __forceinline__ void sum_coherent2(const double *base, int64_t stride, int n, double &re, double &im) {
    re = sum_coherent(base, stride, n);
    im = sum_coherent(base + 256, stride, n);
}

__global__ __launch_bounds__(NTG, 4) void svd_round_fused_kernel_c(const SvdJob *__restrict__ jobs,
                                                                   const BEntry *__restrict__ entries, int round,
                                                                   double2 *__restrict__ W, double2 *__restrict__ G,
                                                                   double *gpart, unsigned int *pair_cnt, unsigned int seq,
                                                                   unsigned int *__restrict__ n_rot,
                                                                   const double *__restrict__ fro2, double rho,
                                                                   int local_sweeps, int full_local, int *err_flag) {
    __shared__ double Xr[NTG / 64][TRJ][CHCP], Xi[NTG / 64][TRJ][CHCP];
    __shared__ double csA[TRJ], cprA[TRJ], cpiA[TRJ];
    __shared__ int partA[TRJ];
    __shared__ int any_flag;
    const BEntry E = entries[blockIdx.x];
    if (E.job < 0) return;
    const SvdJob J = jobs[E.job];
    int64_t bi, bj, NB;
    block_pair_of(J, E.pair, round, bi, bj, NB);
    const int R = (int)J.R, L = (int)J.L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int half = lane >> 5, l31 = lane & 31;      // two rows per load instruction: lanes 0-31 row t, 32-63 row t + 1
    const int row_i = (bi < NB) ? (int)bi * BRJ : R, row_j = (bj < NB) ? (int)bj * BRJ : R;
    const int nchW = (L + CHC - 1) / CHC, nchG = (R + CHC - 1) / CHC;
    const int w_lo = (int)((int64_t)nchW * E.part / E.nparts), w_hi = (int)((int64_t)nchW * (E.part + 1) / E.nparts);
    const int g_lo = (int)((int64_t)nchG * E.part / E.nparts), g_hi = (int)((int64_t)nchG * (E.part + 1) / E.nparts);
    const int nW = w_hi - w_lo, nU = nW + (g_hi - g_lo);
    double (*Xrw)[CHCP] = Xr[wave], (*Xiw)[CHCP] = Xi[wave];
    // S and Q (pitch TRJ + 1 <= CHCP) in the staging planes of wavefronts 2 and 3
    double (*Sr)[TRJ + 1] = reinterpret_cast<double (*)[TRJ + 1]>(&Xr[2][0][0]);
    double (*Si)[TRJ + 1] = reinterpret_cast<double (*)[TRJ + 1]>(&Xi[2][0][0]);
    double (*Qr)[TRJ + 1] = reinterpret_cast<double (*)[TRJ + 1]>(&Xr[3][0][0]);
    double (*Qi)[TRJ + 1] = reinterpret_cast<double (*)[TRJ + 1]>(&Xi[3][0][0]);
    // ---- load the chunks of this part (all in flight): lane (half, l31) holds rows t + half, t = 0, 2, .., 14, column l31
    double2 reg[FITC][TRJ / 2];
#pragma unroll
    for (int it = 0; it < FITC; ++it) {
        const int u = wave + it * (NTG / 64);
        const bool isW = u < nW;
        const double2 *M = isW ? (W + J.w_off) : (G + J.g_off);
        const int len = isW ? L : R;
        const int col = (isW ? (w_lo + u) : (g_lo + (u - nW))) * CHC + l31;
        const bool ok = u < nU && col < len;
#pragma unroll
        for (int t = 0; t < TRJ; t += 2) {
            const int tt = t + half;
            const int row = ((tt < BRJ) ? row_i : row_j) + (tt % BRJ);
            reg[it][t / 2] = (ok && row < R) ? M[(int64_t)row * len + col] : double2{0, 0};
        }
    }
    // ---- partial Gram of the W chunks:  Re S = Xr Xr^T + Xi Xi^T ,  Im S = Xi Xr^T - Xr Xi^T
    d4 ar0 = {0, 0, 0, 0}, ai0 = {0, 0, 0, 0};
#pragma unroll
    for (int it = 0; it < FITC; ++it) {
        const int u = wave + it * (NTG / 64);
        if (u < nW) {   // wave-uniform
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < TRJ; t += 2) {
                Xrw[t + half][l31] = reg[it][t / 2].x;
                Xiw[t + half][l31] = reg[it][t / 2].y;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ks = 0; ks < CHC / 4; ++ks) {
                const double xr = Xrw[l15][ks * 4 + l4], xi = Xiw[l15][ks * 4 + l4];
                ar0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xr, xr, ar0, 0, 0, 0);
                ar0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xi, xi, ar0, 0, 0, 0);
                ai0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xi, xr, ai0, 0, 0, 0);
                ai0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-xr, xi, ai0, 0, 0, 0);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        Xrw[l4 + 4 * r][l15] = ar0[r];
        Xiw[l4 + 4 * r][l15] = ai0[r];
    }
    if (tid == 0) any_flag = 0;
    __syncthreads();
    const int64_t first = (int64_t)blockIdx.x - E.part;
    {
        const int i = tid >> 4, j = tid & 15;
        double sr = 0, si = 0;
#pragma unroll
        for (int w = 0; w < NTG / 64; ++w) {
            sr += Xr[w][i][j];
            si += Xi[w][i][j];
        }
        __hip_atomic_store(&gpart[(int64_t)blockIdx.x * 512 + tid], sr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&gpart[(int64_t)blockIdx.x * 512 + 256 + tid], si, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (E.nparts > 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my stores have reached the coherent level
        __syncthreads();
        if (tid == 0) {
            atomicAdd(&pair_cnt[first], 1u);
            const unsigned int target = (unsigned int)E.nparts * seq;
            unsigned int spins = 0;
            while (__hip_atomic_load(&pair_cnt[first], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > FUSED_SPIN_MAX) {
                    atomicExch(err_flag, 1);
                    break;
                }
            }
        }
    }
    __syncthreads();       // (also: every wavefront is done with the partials in its staging planes)
    {
        double sr, si;
        sum_coherent2(gpart + first * 512 + tid, 512, E.nparts, sr, si);
        Sr[tid >> 4][tid & 15] = sr;
        Si[tid >> 4][tid & 15] = si;
        Qr[tid >> 4][tid & 15] = ((tid >> 4) == (tid & 15)) ? 1.0 : 0.0;
        Qi[tid >> 4][tid & 15] = 0.0;
    }
    __syncthreads();
    const double tol = 2.220446049250313e-16 * sqrt((double)L);
    const double floor2 = svd_floor2(rho, fro2[E.job]);
    {
        const int ei = tid >> 4, ej = tid & 15;
        const bool relevant = full_local ? (ei < ej) : (ei < BRJ && ej >= BRJ);
        const double g2 = Sr[ei][ej] * Sr[ei][ej] + Si[ei][ej] * Si[ei][ej];
        if (relevant && svd_needs_rotation(Sr[ei][ei], Sr[ej][ej], g2, tol, floor2))
            atomicOr(&any_flag, svd_big_rotation(Sr[ei][ei], Sr[ej][ej], g2, floor2) ? 3 : 1);
    }
    __syncthreads();
    if (any_flag == 0) return;
    if (tid == 0 && E.part == 0) {
        atomicAdd(n_rot, 1u);
        if (any_flag & 2) atomicAdd(n_rot + 1, 1u);
    }
    if (wave == 0) svd_local_solve_c(Sr, Si, Qr, Qi, csA, cprA, cpiA, partA, lane, local_sweeps, full_local, tol, floor2);
    __syncthreads();
    // ---- Q to registers BEFORE the staging planes (which hold it) are reused
    double qar[4], qai[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        qar[kk] = Qr[l15][kk * 4 + l4];
        qai[kk] = Qi[l15][kk * 4 + l4];
    }
    __syncthreads();
    // ---- apply X <- Q X (complex): Xr' = Qr Xr - Qi Xi ; Xi' = Qr Xi + Qi Xr
#pragma unroll
    for (int it = 0; it < FITC; ++it) {
        const int u = wave + it * (NTG / 64);
        if (u >= nU) continue;
        const bool isW = u < nW;
        double2 *M = isW ? (W + J.w_off) : (G + J.g_off);
        const int len = isW ? L : R;
        const int c0 = (isW ? (w_lo + u) : (g_lo + (u - nW))) * CHC;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < TRJ; t += 2) {
            Xrw[t + half][l31] = reg[it][t / 2].x;
            Xiw[t + half][l31] = reg[it][t / 2].y;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int tile = 0; tile < CHC / 16; ++tile) {
            d4 orr = {0, 0, 0, 0}, oii = {0, 0, 0, 0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const double br = Xrw[kk * 4 + l4][tile * 16 + l15], bim = Xiw[kk * 4 + l4][tile * 16 + l15];
                orr = __builtin_amdgcn_mfma_f64_16x16x4f64(qar[kk], br, orr, 0, 0, 0);
                orr = __builtin_amdgcn_mfma_f64_16x16x4f64(-qai[kk], bim, orr, 0, 0, 0);
                oii = __builtin_amdgcn_mfma_f64_16x16x4f64(qar[kk], bim, oii, 0, 0, 0);
                oii = __builtin_amdgcn_mfma_f64_16x16x4f64(qai[kk], br, oii, 0, 0, 0);
            }
            const int oc = c0 + tile * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = l4 + 4 * r;
                const int row = ((t < BRJ) ? row_i : row_j) + (t % BRJ);
                if (row < R && oc < len) M[(int64_t)row * len + oc] = double2{orr[r], oii[r]};
            }
        }
    }
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

// Direct Hermitian iteration (tpa_svd_direct): S <- W (the n x n input image, R = L = n), Qtot <- 1; and its epilogue sig_i = S_ii.
template <bool CPLX>
__global__ __launch_bounds__(NT) void eigh_direct_init_kernel(const SvdJob *__restrict__ jobs, const int2 *__restrict__ rows,
                                                              const double *__restrict__ W, double *__restrict__ S, double *__restrict__ Qt) {
    const int gw = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int2 jr = rows[gw];
    if (jr.x < 0) return;
    const SvdJob J = jobs[jr.x];
    const int64_t r = jr.y;
    for (int64_t c = lane; c < J.R; c += 64) {
        if (CPLX) {
            reinterpret_cast<double2 *>(S)[J.g_off + r * J.R + c] = reinterpret_cast<const double2 *>(W)[J.w_off + r * J.L + c];
            reinterpret_cast<double2 *>(Qt)[J.g_off + r * J.R + c] = double2{(c == r) ? 1.0 : 0.0, 0.0};
        } else {
            S[J.g_off + r * J.R + c] = W[J.w_off + r * J.L + c];
            Qt[J.g_off + r * J.R + c] = (c == r) ? 1.0 : 0.0;
        }
    }
}
template <bool CPLX>
__global__ __launch_bounds__(NT) void eigh_direct_diag_kernel(const SvdJob *__restrict__ jobs, const int2 *__restrict__ rows,
                                                              const double *__restrict__ S, double *__restrict__ sig) {
    const int gw = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int2 jr = rows[gw];
    if (jr.x < 0 || (threadIdx.x & 63) != 0) return;
    const SvdJob J = jobs[jr.x];
    sig[J.sig_off + jr.y] = S[(CPLX ? 2 : 1) * (J.g_off + (int64_t)jr.y * J.R + jr.y)];
}

// The order of the singular values on the device (round 5): perm[sig_off + rank(j)] = j with
//     rank(j) = #{i : s_i > s_j} + #{i < j : s_i == s_j},
// the permutation of a stable sort by descending sigma (what the host did with std::stable_sort between two copies: D2H of sigma, a
// synchronisation, the sort, H2D of the permutation; profiles/r05_idle_gap_analysis.txt "svd_norms -> svd_finish").  One wavefront per row
// counts; O(R^2 / 64) compares per wavefront, R <= a few thousand.  A sigma that is not finite sorts last (so that perm is a permutation
// whatever the data) and raises *bad.  Measured: the same permutation (every parity field of the bench line unchanged to the last digit),
// 8.54 instead of 8.54 - 8.62 ms per npc.svd at chi = 2048, 2.70 instead of 2.71 at chi = 512 -- one synchronisation and two copies fewer per
// call, worth less than 1 % (the host sorted while the device had nothing else to do anyway, and the final synchronisation stays).
__global__ __launch_bounds__(NT) void svd_rank_kernel(const SvdJob *__restrict__ jobs, const int2 *__restrict__ rows,
                                                      const double *__restrict__ sig, int64_t *__restrict__ perm, unsigned int *bad) {
    const int gw = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int2 jr = rows[gw];
    if (jr.x < 0) return;
    const SvdJob J = jobs[jr.x];
    const double *s = sig + J.sig_off;
    const int64_t j = jr.y;
    const double sj = s[j];
    const bool fin = isfinite(sj);
    const double kj = fin ? sj : -1.0;
    double before = 0;
    for (int64_t i = lane; i < J.R; i += 64) {
        const double si = s[i];
        const double ki = isfinite(si) ? si : -1.0;
        if (ki > kj || (ki == kj && i < j)) before += 1.0;
    }
    before = wave_sum(before);
    if (lane == 0) {
        perm[J.sig_off + (int64_t)before] = j;
        if (!fin) {
            *bad = 1u;
            __threadfence_system();
        }
    }
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
    const bool tr = (J.tr != 0);
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

// ===================================================================================================
// Rank-revealing preconditioner: Householder QR with column pivoting (BLAS-2, batched over the charge blocks).
//
// DMRG wave-function blocks are numerically rank deficient (rank <= chi of d*chi; measured sigma from 1 down
// to 1e-27 with a cliff) and strongly graded.  Plain one-sided Jacobi then needs ~25 sweeps over ALL rows.
// With X P = Q [R; 0] (X = A or A^T, tall M x N) the Jacobi iteration only has to diagonalise the r x N factor
// R (r = numerical rank): ~7 sweeps over ~half the rows (numpy experiment on real theta blocks: 25 -> 7 sweeps).
//     X = (Q_r U_R) Sigma (VH_R P^T),   SVD(R) = U_R Sigma VH_R  by the block-Jacobi kernels above.
// Per step k two launches for all blocks together: `qrp_pivot_kernel` (one workgroup per block: pivot search on
// the exact residual column norms, column swap, Householder vector; X is kept column-major so that this is coalesced) and `qrp_update_kernel` (column tiles: rank-1
// update of the trailing matrix and exact recomputation of the residual norms in the same pass).
struct QrpJob {  // int64[8]
    int64_t x_off, M, N, c_off, tr, r_off, pad0, pad1;   // c_off: offset into cn / tau / cperm ; tr: X = A^T
};
struct QrpState {  // per job
    int rank, done, nbk, last;   // nbk: size of the current panel; last: that panel was the final one
};

__global__ __launch_bounds__(NT) void qrp_init_kernel(const QrpJob *__restrict__ jobs, const SvdJob *__restrict__ sj,
                                                      const double *__restrict__ A, double *__restrict__ X,
                                                      double *__restrict__ cn, int64_t *__restrict__ cperm,
                                                      QrpState *__restrict__ state) {
    // grid (column tiles of 64, jobs): X = A or A^T stored COLUMN-major (X[j*M + i]), exact column norms^2,
    // identity permutation.  Wave w owns the columns j0 + 16 w .. +15 of the tile, lanes run along the rows
    // (contiguous in X); for X = A the 64 x 64 tile is transposed through LDS so that both sides stay coalesced.
    __shared__ double tile[64][65];
    const QrpJob J = jobs[blockIdx.y];
    const SvdJob S = sj[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t j0 = (int64_t)blockIdx.x * 64;
    if (j0 >= J.N) return;
    double acc[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) acc[rr] = 0.0;
    for (int64_t i0 = 0; i0 < J.M; i0 += 64) {
        if (!J.tr) {
            __syncthreads();
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int64_t i = i0 + wave * 16 + rr, j = j0 + lane;
                tile[wave * 16 + rr][lane] = (i < J.M && j < J.N) ? A[S.a_off + i * S.n + j] : 0.0;
            }
            __syncthreads();
        }
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int64_t j = j0 + wave * 16 + rr, i = i0 + lane;
            if (j < J.N && i < J.M) {
                const double v = J.tr ? A[S.a_off + j * S.n + i] : tile[lane][wave * 16 + rr];
                X[J.x_off + j * J.M + i] = v;
                acc[rr] = fma(v, v, acc[rr]);
            }
        }
    }
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
        const double t = wave_sum(acc[rr]);
        const int64_t j = j0 + wave * 16 + rr;
        if (lane == 0 && j < J.N) {
            cn[J.c_off + j] = t;
            cperm[J.c_off + j] = j;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) state[blockIdx.y] = QrpState{0, 0, 0, 0};
}

constexpr int NTP_MAX = 256;   // threads of the panel kernel: 64 (one wavefront, no cross-wave barriers) or 256
constexpr int PNB = 8;    // pivot columns factorised per launch (panel pivoting: the PNB largest residual columns)
constexpr int RPT_MAX = 32;  // rows / candidate columns per thread held in registers (template RPT = 8, 16, 32)  ->  max(m, n) <= 8192

// sum the values v[q], q < n or q == extra, over the workgroup (NTP threads); results valid in every thread.  The callers
// sit in fully unrolled loops, so n / extra are constants after unrolling and the unused entries cost nothing.
template <int NTP, int K>
__device__ __forceinline__ void block_sum_vec(double (&v)[K], double (*red)[PNB + 1], int n, int extra) {
#pragma unroll
    for (int q = 0; q < K; ++q)
        if (q < n || q == extra) v[q] = wave_sum(v[q]);
    if (NTP == 64) return;   // a single wavefront: wave_sum already left the total in every lane
    lds_barrier();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < K; ++q)
            if (q < n || q == extra) red[threadIdx.x >> 6][q] = v[q];
    }
    lds_barrier();
#pragma unroll
    for (int q = 0; q < K; ++q)
        if (q < n || q == extra) {
            double t = 0;
#pragma unroll
            for (int w = 0; w < NTP / 64; ++w) t += red[w][q];
            v[q] = t;
        }
}

// One workgroup per block: pick the (up to) PNB unprocessed columns with the largest residual norms and factorise
// that panel (Householder vectors -> Vall, R entries -> X, compact-WY factor -> Tpan).  Columns are never moved:
// cperm[k + l] records which physical column became logical column k + l, cn[j] = -1 marks column j as done, and the
// trailing update walks over the physical columns skipping the marked ones.  Greedy pivoting is exact for the first
// column of a panel and by pre-panel norms for the others; the rank decision is unaffected because the trailing update
// recomputes every residual norm exactly.
// (A one-wavefront variant with the whole panel in the registers of one SIMD was 1.6x slower: the per-lane serial work
// outweighs the saved barriers.)
template <int NTP, int RPT>
__global__ __launch_bounds__(NTP) void qrp_panel_kernel(const QrpJob *__restrict__ jobs, int k, double *__restrict__ X,
                                                       double *__restrict__ Vall, double *__restrict__ cn,
                                                       double *__restrict__ tau, int64_t *__restrict__ cperm,
                                                       QrpState *__restrict__ state, const double *__restrict__ fro2,
                                                       double tol2, double *__restrict__ Tpan, int pivot) {
    __shared__ double rv[2 * (NTP / 64)];
    __shared__ int64_t ri[2 * (NTP / 64)];
    __shared__ double red[NTP / 64][PNB + 1];
    __shared__ int64_t s_p[PNB];
    __shared__ int s_nbk;
    __shared__ double s_alpha, s_vrow[PNB], Tf[PNB][PNB];
    const int b = blockIdx.x;
    const QrpJob J = jobs[b];
    const QrpState st0 = state[b];
    if (st0.done) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t M = J.M, N = J.N;
    const int64_t Kmax = (M < N) ? M : N;      // number of reflectors (N <= M in the pivoted use)
    if (st0.last || k >= Kmax) {
        if (tid == 0) state[b] = QrpState{st0.last ? st0.rank : (int)Kmax, 1, 0, 0};
        return;
    }
    // ---- the PNB largest residual norms among the unprocessed columns (ties -> smallest index: deterministic)
    double cand[RPT];
#pragma unroll
    for (int t = 0; t < RPT; ++t) {
        const int64_t j = tid + (int64_t)t * NTP;
        cand[t] = (j < N) ? cn[J.c_off + j] : -4.0;       // processed columns hold -1
    }
    const double thresh = tol2 * fro2[b];
    if (tid == 0) s_nbk = 0;
    if (!pivot) {   // plain QR: the next PNB columns in their natural order, no rank test
        if (tid == 0) {
            const int nb_ = (int)((Kmax - k < PNB) ? (Kmax - k) : PNB);
            for (int l = 0; l < nb_; ++l) s_p[l] = (int64_t)k + l;
            s_nbk = nb_;
        }
        lds_barrier();
    } else {
        // One barrier per pivot: every wavefront finds its best candidate (value by wave_max, smallest index among equal values
        // by wave_min), publishes it in a double-buffered LDS slot, and EVERY thread merges the NTP/64 entries itself.
        int nsel = 0;
        for (int l = 0; l < PNB; ++l) {
            double bv = -3.0;
            int bidx = 0x7fffffff;
#pragma unroll
            for (int t = 0; t < RPT; ++t)
                if (cand[t] > bv) {
                    bv = cand[t];
                    bidx = tid + t * NTP;
                }
            const double wv = wave_max(bv);
            const int wi = wave_min((bv == wv) ? bidx : 0x7fffffff);
            if (lane == 0) {
                rv[(l & 1) * (NTP / 64) + wave] = wv;
                ri[(l & 1) * (NTP / 64) + wave] = wi;
            }
            lds_barrier();
            double v0 = rv[(l & 1) * (NTP / 64)];
            int64_t i0 = ri[(l & 1) * (NTP / 64)];
#pragma unroll
            for (int w = 1; w < NTP / 64; ++w) {
                const double vw = rv[(l & 1) * (NTP / 64) + w];
                const int64_t iw = ri[(l & 1) * (NTP / 64) + w];
                if (vw > v0 || (vw == v0 && iw < i0)) {
                    v0 = vw;
                    i0 = iw;
                }
            }
            if (!(v0 > thresh)) break;   // uniform: every thread merged the same entries
            if (tid == 0) s_p[l] = i0;
            nsel = l + 1;
#pragma unroll
            for (int t = 0; t < RPT; ++t)
                if (tid + t * NTP == (int)i0) cand[t] = -4.0;
        }
        if (tid == 0) s_nbk = nsel;
        lds_barrier();
    }
    const int nbk = s_nbk;
    if (nbk == 0) {
        if (tid == 0) state[b] = QrpState{k, 1, 0, 0};
        return;
    }
    if (tid == 0) {
        const bool last = (nbk < PNB);
        state[b] = QrpState{last ? k + nbk : 0, 0, nbk, last ? 1 : 0};
        for (int x = 0; x < PNB; ++x)
            for (int y = 0; y < PNB; ++y) Tf[x][y] = 0.0;
    }
    if (tid < nbk) {
        cperm[J.c_off + k + tid] = s_p[tid];
        cn[J.c_off + s_p[tid]] = -1.0;     // processed
    }
    lds_barrier();
    double *Xb = X + J.x_off;
    // ---- gather the panel into registers (X is column-major: contiguous loads)
    double c[PNB][RPT];
#pragma unroll
    for (int t = 0; t < RPT; ++t) {
        const int64_t i = tid + (int64_t)t * NTP;
#pragma unroll
        for (int l = 0; l < PNB; ++l) c[l][t] = (i < M && l < nbk) ? Xb[s_p[l] * M + i] : 0.0;
    }
    // ---- factorise the panel, column by column (must be fully unrolled: c[l] has to stay in registers)
#pragma clang loop unroll(full)
    for (int l = 0; l < PNB; ++l) {
        if (l < nbk) {   // uniform
            const int64_t kl = (int64_t)k + l;
            if (l > 0) {
                // c_l <- (I - V Tf^T V^T) c_l  with the l reflectors found so far (their vectors sit in c[0..l-1])
                double y[PNB];
#pragma unroll
                for (int m = 0; m < PNB; ++m) {
                    y[m] = 0.0;
                    if (m < l) {
#pragma unroll
                        for (int t = 0; t < RPT; ++t) y[m] = fma(c[m][t], c[l][t], y[m]);
                    }
                }
                block_sum_vec<NTP, PNB>(y, red, l, -1);
                double z[PNB];
#pragma unroll
                for (int m = 0; m < PNB; ++m) {
                    z[m] = 0.0;
                    if (m < l) {
#pragma unroll
                        for (int mm = 0; mm < PNB; ++mm)
                            if (mm <= m) z[m] = fma(Tf[mm][m], y[mm], z[m]);
                    }
                }
#pragma unroll
                for (int m = 0; m < PNB; ++m)
                    if (m < l) {
#pragma unroll
                        for (int t = 0; t < RPT; ++t) c[l][t] = fma(-c[m][t], z[m], c[l][t]);
                    }
            }
            // norm below the diagonal and inner products with the earlier vectors (for Tf) in one reduction; the diagonal
            // element alpha and row kl of the earlier vectors are broadcast through LDS by the thread that owns row kl
            double g[PNB + 1];
#pragma unroll
            for (int q = 0; q <= PNB; ++q) g[q] = 0.0;
#pragma unroll
            for (int t = 0; t < RPT; ++t) {
                const int64_t i = tid + (int64_t)t * NTP;
                if (i > kl && i < M) {
                    g[PNB] = fma(c[l][t], c[l][t], g[PNB]);
#pragma unroll
                    for (int m = 0; m < PNB; ++m)
                        if (m < l) g[m] = fma(c[m][t], c[l][t], g[m]);
                } else if (i == kl) {
                    s_alpha = c[l][t];
#pragma unroll
                    for (int m = 0; m < PNB; ++m)
                        if (m < l) s_vrow[m] = c[m][t];
                }
            }
            block_sum_vec<NTP, PNB + 1>(g, red, l, PNB);
            const double s2 = g[PNB], alpha = s_alpha;
            // beta = -sign(alpha) |x|,  tau = (beta - alpha) / beta = 1 + |alpha| / |x|,  scale = 1 / (alpha - beta): one reciprocal
            // square root and one reciprocal, both from the hardware seed + Newton steps (~1 ulp; a Householder vector does not need
            // correctly rounded divisions, and these sit on the serial path of every column)
            double beta = alpha, tk = 0.0, scale = 0.0;
            if (s2 > 0.0) {
                const double x2 = fma(alpha, alpha, s2);
                double r = __builtin_amdgcn_rsq(x2);
                r = r * fma(-0.5 * x2 * r, r, 1.5);
                r = r * fma(-0.5 * x2 * r, r, 1.5);          // 1 / |x|
                const double nx = x2 * r;                   // |x|
                beta = -copysign(nx, alpha);
                tk = fma(fabs(alpha), r, 1.0);
                const double d = alpha - beta;              // = sign(alpha) (|alpha| + |x|): no cancellation
                double q = __builtin_amdgcn_rcp(d);
                q = q * fma(-d, q, 2.0);
                scale = q * fma(-d, q, 2.0);
            }
            // column l of the compact-WY factor: Tf[i2][l] = -tau sum_{m=i2}^{l-1} Tf[i2][m] (v_m^T v_l), thread i2 each
            if (tid < l) {
                double acc = 0;
#pragma unroll
                for (int m = 0; m < PNB; ++m)
                    if (m >= tid && m < l) acc = fma(Tf[tid][m], g[m] * scale + s_vrow[m], acc);
                Tf[tid][l] = -tk * acc;
            } else if (tid == l) {
                Tf[l][l] = tk;
                tau[J.c_off + kl] = tk;
            }
            double *vk = Vall + J.x_off + kl * M;
            double *xc = Xb + s_p[l] * M;
#pragma unroll
            for (int t = 0; t < RPT; ++t) {
                const int64_t i = tid + (int64_t)t * NTP;
                if (i < M) {   // (selects, no divergent branches)
                    const double cv = c[l][t];
                    const double v = (i > kl) ? cv * scale : ((i == kl) ? 1.0 : 0.0);
                    xc[i] = (i < kl) ? cv : ((i == kl) ? beta : 0.0);   // R entry (rows k .. kl-1 were changed by this panel's earlier reflectors)
                    vk[i] = v;
                    c[l][t] = v;
                }
            }
            lds_barrier();   // Tf column l, s_alpha / s_vrow reuse
        }
    }
    if (tid < PNB * PNB) Tpan[(int64_t)b * PNB * PNB + tid] = Tf[tid / PNB][tid % PNB];
}

// ---- compact-WY block reflector applied to a 16-column tile of a row-major matrix, on the matrix cores ----------
//     C[k0:, tile] <- (I - V Tf' V^T) C[k0:, tile],    Tf' = Tf^T (TRANS: trailing update of the factorisation)
//                                                      or Tf (forming Q U_R),
// V[l*M + i] = component i of reflector l (0 above its diagonal), nb <= NB reflectors, Tf upper triangular in LDS.
// 1024 threads = 16 wavefronts, each owning 16-row slabs (slab s of wave w = rows k0 + 16 (w + 16 s) ...):
//   pass 1  Y  = V^T C      v_mfma_f64_16x16x4: A(l, i) = V, B(i, j) = C, K runs over the rows      -> LDS reduce
//   small   Z  = Tf' Y      (NB x 16, 256 threads)
//   pass 2  C -= V Z        A(i, l) = V, B(l, j) = -Z, accumulator preloaded with the C slab (NB/4 MFMAs per slab)
// Returns (NORMS) sum_{i >= kend} C[i, j]^2 of the updated tile column j = j0 + (lane & 15), valid in threads < 16.
constexpr int NTR = 1024, RCOLS = 16;

template <int NB, int NTH = NTR>
struct WySmem {
    double red[NTH / 64][NB][RCOLS];
    double Y[NB][RCOLS], Z[NB][RCOLS], T[NB][NB];
    double nrm[NTH / 64][RCOLS];
};

template <int NB, bool TRANS, bool NORMS, int NTH = NTR>
__device__ __forceinline__ double wy_apply_tile(double *Cb, int64_t rs, int64_t cs, int64_t k0, int64_t M, int64_t j0, int64_t jend,
                                                const double *__restrict__ V, int nb, int64_t kend, WySmem<NB, NTH> &sm,
                                                bool col_ok = true) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lo = lane & 15, kq = lane >> 4;
    const int64_t j = j0 + lo;
    const bool jok = (j < jend) && col_ok;
    // ---- pass 1: Y(l, j) = sum_i V(l, i) C(i, j)
    d4 acc = {0, 0, 0, 0};
    for (int64_t i0 = k0 + 16 * wave; i0 < M; i0 += 16 * (NTH / 64)) {
        double a[4], bb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t i = i0 + 4 * kq + q;
            const bool iok = i < M;
            a[q] = (iok && lo < nb) ? V[(int64_t)lo * M + i] : 0.0;
            bb[q] = (iok && jok) ? Cb[i * rs + j * cs] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], bb[q], acc, 0, 0, 0);
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
        if (kq + 4 * reg < NB) sm.red[wave][kq + 4 * reg][lo] = acc[reg];
    __syncthreads();
    if (threadIdx.x < NB * RCOLS) {
        const int l = threadIdx.x >> 4, c = threadIdx.x & 15;
        double t = 0;
#pragma unroll
        for (int q = 0; q < NTH / 64; ++q) t += sm.red[q][l][c];
        sm.Y[l][c] = t;
    }
    __syncthreads();
    if (threadIdx.x < NB * RCOLS) {
        const int l = threadIdx.x >> 4, c = threadIdx.x & 15;
        double t = 0;
#pragma unroll
        for (int m = 0; m < NB; ++m) t = fma(TRANS ? sm.T[m][l] : sm.T[l][m], sm.Y[m][c], t);
        sm.Z[l][c] = -t;
    }
    __syncthreads();
    double zneg[NB / 4];
#pragma unroll
    for (int q = 0; q < NB / 4; ++q) zneg[q] = sm.Z[4 * q + kq][lo];
    // ---- pass 2: two 16-row slabs per iteration (all loads before the stores: the compiler cannot prove that the
    //      stores of one slab do not alias the loads of the next)
    double nrm = 0;
    for (int64_t i0 = k0 + 16 * wave; i0 < M; i0 += 2 * 16 * (NTH / 64)) {
        d4 c[2];
        double av[2][NB / 4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t ib = i0 + (int64_t)h * 16 * (NTH / 64);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int64_t i = ib + kq + 4 * reg;
                c[h][reg] = (i < M && jok) ? Cb[i * rs + j * cs] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < NB / 4; ++q) {
                const int64_t i = ib + lo;
                const int l = 4 * q + kq;
                av[h][q] = (i < M && l < nb) ? V[(int64_t)l * M + i] : 0.0;
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < NB / 4; ++q) c[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[h][q], zneg[q], c[h], 0, 0, 0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t ib = i0 + (int64_t)h * 16 * (NTH / 64);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int64_t i = ib + kq + 4 * reg;
                if (i < M && jok) {
                    Cb[i * rs + j * cs] = c[h][reg];
                    if (NORMS && i >= kend) nrm = fma(c[h][reg], c[h][reg], nrm);
                }
            }
        }
    }
    if (!NORMS) return 0.0;
    nrm += __shfl_xor(nrm, 16, 64);
    nrm += __shfl_xor(nrm, 32, 64);
    if (lane < RCOLS) sm.nrm[wave][lo] = nrm;
    __syncthreads();
    double t = 0;
    if (threadIdx.x < RCOLS) {
#pragma unroll
        for (int q = 0; q < NTH / 64; ++q) t += sm.nrm[q][threadIdx.x];
    }
    return t;
}

// trailing update with the panel's nbk reflectors:  X[k:, j] <- (I - V Tf^T V^T) X[k:, j]  for every column j that has not
// been factorised yet (cn[j] >= 0; tiles walk over the physical columns), plus the exact residual norms (rows >= k + nbk).
__global__ __launch_bounds__(NTR) void qrp_update_kernel(const QrpJob *__restrict__ jobs, int k, double *__restrict__ X,
                                                         const double *__restrict__ Vall, double *__restrict__ cn,
                                                         const QrpState *__restrict__ state,
                                                         const double *__restrict__ Tpan) {
    __shared__ WySmem<PNB> sm;
    const int b = blockIdx.y;
    const QrpState st = state[b];
    if (st.done || st.nbk == 0) return;
    const QrpJob J = jobs[b];
    const int nbk = st.nbk;
    const int64_t j0 = (int64_t)blockIdx.x * RCOLS;
    if (j0 >= J.N) return;
    const int64_t jl = j0 + (threadIdx.x & (RCOLS - 1));
    const bool col_ok = (jl < J.N) && (cn[J.c_off + jl] >= 0.0);
    if (__ballot(col_ok) == 0) return;   // same 16 columns in every wavefront: uniform over the workgroup
    if (threadIdx.x < PNB * PNB) sm.T[threadIdx.x / PNB][threadIdx.x % PNB] = Tpan[(int64_t)b * PNB * PNB + threadIdx.x];
    const double nrm = wy_apply_tile<PNB, true, true>(X + J.x_off, 1, J.M, k, J.M, j0, J.N, Vall + J.x_off + (int64_t)k * J.M,
                                                      nbk, (int64_t)k + nbk, sm, col_ok);
    if (threadIdx.x < RCOLS && col_ok) cn[J.c_off + jl] = nrm;
}

// after the factorisation: the columns that were never used as pivots become the logical columns r .. N-1 (in
// increasing physical order)
__global__ __launch_bounds__(NT) void qrp_finish_perm_kernel(const QrpJob *__restrict__ jobs, const QrpState *__restrict__ state,
                                                             const double *__restrict__ cn, int64_t *__restrict__ cperm) {
    __shared__ int cnt[NT];
    const QrpJob J = jobs[blockIdx.x];
    const int64_t N = J.N, r = state[blockIdx.x].rank;
    const int64_t per = (N + NT - 1) / NT, lo = (int64_t)threadIdx.x * per, hi = (lo + per < N) ? lo + per : N;
    int mine = 0;
    for (int64_t j = lo; j < hi; ++j) mine += (cn[J.c_off + j] >= 0.0) ? 1 : 0;
    cnt[threadIdx.x] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int t = 0; t < NT; ++t) {
            const int c = cnt[t];
            cnt[t] = run;
            run += c;
        }
    }
    __syncthreads();
    int64_t pos = r + cnt[threadIdx.x];
    for (int64_t j = lo; j < hi; ++j)
        if (cn[J.c_off + j] >= 0.0) cperm[J.c_off + pos++] = j;
}

// R_top (r x N, contiguous, logical column order) = upper-trapezoidal part of the first r rows of X
__global__ __launch_bounds__(NT) void qrp_extract_kernel(const QrpJob *__restrict__ jobs, const QrpState *__restrict__ state,
                                                         const double *__restrict__ X, const int64_t *__restrict__ cperm,
                                                         double *__restrict__ Rtop) {
    const QrpJob J = jobs[blockIdx.y];
    const int64_t r = state[blockIdx.y].rank, N = J.N;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < r * N; e += (int64_t)gridDim.x * NT) {
        const int64_t i = e / N, j = e - i * N;
        Rtop[J.r_off + e] = (j >= i) ? X[J.x_off + cperm[J.c_off + j] * J.M + i] : 0.0;
    }
}

// T (M x r, row-major, stored at x_off) = [U_R ; 0]
__global__ __launch_bounds__(NT) void qrp_form_t_kernel(const QrpJob *__restrict__ jobs, const QrpState *__restrict__ state,
                                                        const double *__restrict__ UR, double *__restrict__ T) {
    const QrpJob J = jobs[blockIdx.y];
    const int64_t r = state[blockIdx.y].rank, M = J.M;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < M * r; e += (int64_t)gridDim.x * NT) {
        const int64_t i = e / r;
        T[J.x_off + e] = (i < r) ? UR[J.r_off + e] : 0.0;
    }
}

// ---- blocked application of Q_r = H_0 ... H_{r-1} (compact WY, 16 reflectors per launch) ------------------------
// H_{k0} ... H_{k0+15} = I - V Tf V^T  with Tf upper triangular (LAPACK dlarft, forward / columnwise).
constexpr int QNB = 16;

// one workgroup per (block of 16 reflectors, job): Gram of the panel rows, then the Tf recurrence.  One launch
// covers every block of every job.
__global__ __launch_bounds__(NTR) void qrp_tfactor_kernel(const QrpJob *__restrict__ jobs, const QrpState *__restrict__ state,
                                                          const double *__restrict__ Vall, const double *__restrict__ tau,
                                                          double *__restrict__ Tfac) {
    __shared__ double S[QNB][QNB + 1];
    __shared__ double Tf[QNB][QNB + 1];
    const QrpJob J = jobs[blockIdx.y];
    const int64_t r = state[blockIdx.y].rank, M = J.M;
    const int64_t k0 = (int64_t)blockIdx.x * QNB;
    if (k0 >= r) return;
    const int nb = (int)((r - k0 < QNB) ? (r - k0) : QNB);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double *V = Vall + J.x_off + k0 * M;
    // 16 waves x 16 (l, l') pairs each: pair index = wave * 16 + q  ->  l = wave, l' = q
    for (int q = 0; q < QNB; ++q) {
        const int l = wave, lp = q;
        double acc = 0;
        if (l < lp && lp < nb)
            for (int64_t i = k0 + lp + lane; i < M; i += 64) acc = fma(V[l * M + i], V[lp * M + i], acc);
        acc = wave_sum(acc);
        if (lane == 0) S[l][lp] = acc;
    }
    __syncthreads();
    if (threadIdx.x < QNB * QNB) Tf[threadIdx.x >> 4][threadIdx.x & 15] = 0.0;
    __syncthreads();
    for (int j = 0; j < nb; ++j) {
        const double tj = tau[J.c_off + k0 + j];
        if (threadIdx.x < j) {
            const int i = threadIdx.x;
            double acc = 0;
            for (int m = i; m < j; ++m) acc = fma(Tf[i][m], S[m][j], acc);
            Tf[i][j] = -tj * acc;
        } else if (threadIdx.x == j)
            Tf[j][j] = tj;
        __syncthreads();
    }
    if (threadIdx.x < QNB * QNB)
        Tfac[(J.pad0 + blockIdx.x) * (QNB * QNB) + threadIdx.x] = Tf[threadIdx.x >> 4][threadIdx.x & 15];
}

// C[k0:, tile] <- (I - V Tf V^T) C[k0:, tile]   for one block of 16 reflectors and a 16-column tile of C (M x r)
__global__ __launch_bounds__(NTR) void qrp_apply_q_block_kernel(const QrpJob *__restrict__ jobs,
                                                                const QrpState *__restrict__ state, int blk,
                                                                double *__restrict__ C, const double *__restrict__ Vall,
                                                                const double *__restrict__ Tfac) {
    __shared__ WySmem<QNB> sm;
    const int b = blockIdx.y;
    const QrpJob J = jobs[b];
    const int64_t r = state[b].rank;
    const int64_t k0 = (int64_t)blk * QNB;
    if (k0 >= r) return;
    const int64_t j0 = (int64_t)blockIdx.x * RCOLS;
    if (j0 >= r) return;
    const int nb = (int)((r - k0 < QNB) ? (r - k0) : QNB);
    if (threadIdx.x < QNB * QNB) sm.T[threadIdx.x >> 4][threadIdx.x & 15] = Tfac[(J.pad0 + blk) * (QNB * QNB) + threadIdx.x];
    wy_apply_tile<QNB, false, false>(C + J.x_off, r, 1, k0, J.M, j0, r, Vall + J.x_off + k0 * J.M, nb, 0, sm);
}

// final outputs from T = Q_r U_R (M x r), S_R, VH_R (r x N) and the column permutation
__global__ __launch_bounds__(NT) void qrp_output_kernel(const QrpJob *__restrict__ jobs, const SvdJob *__restrict__ sj,
                                                        const QrpState *__restrict__ state, const double *__restrict__ T,
                                                        const double *__restrict__ SR, const double *__restrict__ VHR,
                                                        const int64_t *__restrict__ cperm, double *__restrict__ U,
                                                        double *__restrict__ S, double *__restrict__ VH) {
    const QrpJob J = jobs[blockIdx.y];
    const SvdJob O = sj[blockIdx.y];
    const int64_t r = state[blockIdx.y].rank, M = J.M, N = J.N, K = N;  // K = min(m, n)
    const int64_t stride = (int64_t)gridDim.x * NT, t0 = (int64_t)blockIdx.x * NT + threadIdx.x;
    for (int64_t e = t0; e < K; e += stride) S[O.s_off + e] = (e < r) ? SR[J.c_off + e] : 0.0;
    if (!J.tr) {
        // A = X (m = M >= n = N):  U = T (M x K, zero padded),  VH[jj][cperm[c]] = VH_R[jj][c]
        for (int64_t e = t0; e < M * K; e += stride) {
            const int64_t i = e / K, jj = e - i * K;
            U[O.u_off + e] = (jj < r) ? T[J.x_off + i * r + jj] : 0.0;
        }
        for (int64_t e = t0; e < K * N; e += stride) {
            const int64_t jj = e / N, c = e - jj * N;
            VH[O.vh_off + jj * N + cperm[J.c_off + c]] = (jj < r) ? VHR[J.r_off + jj * N + c] : 0.0;
        }
    } else {
        // A = X^T (m = N < n = M):  U[cperm[c]][jj] = VH_R[jj][c],  VH[jj][c] = T[c][jj]
        for (int64_t e = t0; e < N * K; e += stride) {
            const int64_t c = e / K, jj = e - c * K;
            U[O.u_off + cperm[J.c_off + c] * K + jj] = (jj < r) ? VHR[J.r_off + jj * N + c] : 0.0;
        }
        for (int64_t e = t0; e < K * M; e += stride) {
            const int64_t jj = e / M, c = e - jj * M;
            VH[O.vh_off + e] = (jj < r) ? T[J.x_off + c * r + jj] : 0.0;
        }
    }
}

// ===================================================================================================
// Complex version of the rank-revealing preconditioner (same structure; interleaved double2 storage).
//   X = A (m >= n) or A^H (m < n), column-major;   X P = Q [R; 0],   H = I - tau v v^H (LAPACK zlarfg: beta real),
//   the factorisation applies H^H from the left, blocks H_0 .. H_{l-1} = I - V T V^H (zlarft, forward / columnwise).
//   A = X:    U = Q_r U_R,          VH[jj][cperm[c]] = VH_R[jj][c]
//   A = X^H:  U[cperm[c]][jj] = conj(VH_R[jj][c]),   VH[jj][c] = conj((Q_r U_R)[c][jj])
// Complex products on the matrix cores are 4 real MFMAs on the (re, im) planes of the fragments.
typedef double2 cd;
__device__ __forceinline__ cd c_mul(cd a, cd b) { return cd{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cd c_mulc(cd a, cd b) { return cd{a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x}; }   // conj(a) * b
__device__ __forceinline__ cd c_fma(cd a, cd b, cd acc) { return cd{acc.x + a.x * b.x - a.y * b.y, acc.y + a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cd c_fmac(cd a, cd b, cd acc) { return cd{acc.x + a.x * b.x + a.y * b.y, acc.y + a.x * b.y - a.y * b.x}; }  // acc + conj(a) b

__global__ __launch_bounds__(NT) void qrp_init_kernel_c(const QrpJob *__restrict__ jobs, const SvdJob *__restrict__ sj,
                                                        const cd *__restrict__ A, cd *__restrict__ X,
                                                        double *__restrict__ cn, int64_t *__restrict__ cperm,
                                                        QrpState *__restrict__ state) {
    // grid (column tiles of 64, jobs); wave w owns the columns j0 + 16 w .. +15, lanes run along the rows
    __shared__ double tr_[64][65], ti_[64][65];
    const QrpJob J = jobs[blockIdx.y];
    const SvdJob S = sj[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t j0 = (int64_t)blockIdx.x * 64;
    if (j0 >= J.N) return;
    double acc[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) acc[rr] = 0.0;
    for (int64_t i0 = 0; i0 < J.M; i0 += 64) {
        if (!J.tr) {
            __syncthreads();
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int64_t i = i0 + wave * 16 + rr, j = j0 + lane;
                const cd v = (i < J.M && j < J.N) ? A[S.a_off + i * S.n + j] : cd{0.0, 0.0};
                tr_[wave * 16 + rr][lane] = v.x;
                ti_[wave * 16 + rr][lane] = v.y;
            }
            __syncthreads();
        }
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int64_t j = j0 + wave * 16 + rr, i = i0 + lane;
            if (j < J.N && i < J.M) {
                cd v;
                if (J.tr) {          // X = A^H:  X[i][j] = conj(A[j][i])
                    v = A[S.a_off + j * S.n + i];
                    v.y = -v.y;
                } else
                    v = cd{tr_[lane][wave * 16 + rr], ti_[lane][wave * 16 + rr]};
                X[J.x_off + j * J.M + i] = v;
                acc[rr] = fma(v.x, v.x, fma(v.y, v.y, acc[rr]));
            }
        }
    }
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
        const double t = wave_sum(acc[rr]);
        const int64_t j = j0 + wave * 16 + rr;
        if (lane == 0 && j < J.N) {
            cn[J.c_off + j] = t;
            cperm[J.c_off + j] = j;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) state[blockIdx.y] = QrpState{0, 0, 0, 0};
}

constexpr int KC = 2 * PNB + 1;   // real values reduced together in the complex panel kernel

template <int NTP, int K>
__device__ __forceinline__ void block_sum_arr(double (&v)[K], double (*red)[KC]) {
#pragma unroll
    for (int q = 0; q < K; ++q) v[q] = wave_sum(v[q]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < K; ++q) red[threadIdx.x >> 6][q] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < K; ++q) {
        double t = 0;
#pragma unroll
        for (int w = 0; w < NTP / 64; ++w) t += red[w][q];
        v[q] = t;
    }
}

template <int NTP, int RPT>
__global__ __launch_bounds__(NTP) void qrp_panel_kernel_c(const QrpJob *__restrict__ jobs, int k, cd *__restrict__ X,
                                                          cd *__restrict__ Vall, double *__restrict__ cn,
                                                          cd *__restrict__ tau, int64_t *__restrict__ cperm,
                                                          QrpState *__restrict__ state, const double *__restrict__ fro2,
                                                          double tol2, cd *__restrict__ Tpan, int pivot) {
    __shared__ double rv[NTP / 64];
    __shared__ int64_t ri[NTP / 64];
    __shared__ double red[NTP / 64][KC];
    __shared__ int64_t s_p[PNB];
    __shared__ int s_nbk;
    __shared__ cd s_alpha, s_vrow[PNB], Tf[PNB][PNB];
    const int b = blockIdx.x;
    const QrpJob J = jobs[b];
    const QrpState st0 = state[b];
    if (st0.done) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t M = J.M, N = J.N;
    const int64_t Kmax = (M < N) ? M : N;      // number of reflectors (N <= M in the pivoted use)
    if (st0.last || k >= Kmax) {
        if (tid == 0) state[b] = QrpState{st0.last ? st0.rank : (int)Kmax, 1, 0, 0};
        return;
    }
    double cand[RPT];
#pragma unroll
    for (int t = 0; t < RPT; ++t) {
        const int64_t j = tid + (int64_t)t * NTP;
        cand[t] = (j < N) ? cn[J.c_off + j] : -4.0;
    }
    const double thresh = tol2 * fro2[b];
    if (tid == 0) s_nbk = 0;
    if (!pivot) {   // plain QR: the next PNB columns in their natural order, no rank test
        if (tid == 0) {
            const int nb_ = (int)((Kmax - k < PNB) ? (Kmax - k) : PNB);
            for (int l = 0; l < nb_; ++l) s_p[l] = (int64_t)k + l;
            s_nbk = nb_;
        }
        __syncthreads();
    } else
    for (int l = 0; l < PNB; ++l) {
        double bv = -3.0;
        int64_t bidx = N;
#pragma unroll
        for (int t = 0; t < RPT; ++t)
            if (cand[t] > bv) {
                bv = cand[t];
                bidx = tid + (int64_t)t * NTP;
            }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(bv, off, 64);
            const int64_t oi = __shfl_xor(bidx, off, 64);
            if (ov > bv || (ov == bv && oi < bidx)) {
                bv = ov;
                bidx = oi;
            }
        }
        __syncthreads();
        if (lane == 0) {
            rv[wave] = bv;
            ri[wave] = bidx;
        }
        __syncthreads();
        if (tid == 0) {
            double v0 = rv[0];
            int64_t i0 = ri[0];
            for (int w = 1; w < NTP / 64; ++w)
                if (rv[w] > v0 || (rv[w] == v0 && ri[w] < i0)) {
                    v0 = rv[w];
                    i0 = ri[w];
                }
            if (v0 > thresh && s_nbk == l) {
                s_p[l] = i0;
                s_nbk = l + 1;
            }
        }
        __syncthreads();
        if (s_nbk <= l) break;
        const int64_t w = s_p[l];
#pragma unroll
        for (int t = 0; t < RPT; ++t)
            if (tid + (int64_t)t * NTP == w) cand[t] = -4.0;
    }
    const int nbk = s_nbk;
    if (nbk == 0) {
        if (tid == 0) state[b] = QrpState{k, 1, 0, 0};
        return;
    }
    if (tid == 0) {
        const bool last = (nbk < PNB);
        state[b] = QrpState{last ? k + nbk : 0, 0, nbk, last ? 1 : 0};
        for (int x = 0; x < PNB; ++x)
            for (int y = 0; y < PNB; ++y) Tf[x][y] = cd{0.0, 0.0};
    }
    if (tid < nbk) {
        cperm[J.c_off + k + tid] = s_p[tid];
        cn[J.c_off + s_p[tid]] = -1.0;
    }
    __syncthreads();
    cd *Xb = X + J.x_off;
    cd c[PNB][RPT];
#pragma unroll
    for (int t = 0; t < RPT; ++t) {
        const int64_t i = tid + (int64_t)t * NTP;
#pragma unroll
        for (int l = 0; l < PNB; ++l) c[l][t] = (i < M && l < nbk) ? Xb[s_p[l] * M + i] : cd{0.0, 0.0};
    }
#pragma clang loop unroll(full)
    for (int l = 0; l < PNB; ++l) {
        if (l < nbk) {
            const int64_t kl = (int64_t)k + l;
            if (l > 0) {
                // c_l <- (I - V T^H V^H) c_l :  y = V^H c_l,  z = T^H y,  c_l -= V z
                double y[2 * PNB];
#pragma unroll
                for (int m = 0; m < PNB; ++m) {
                    cd a{0.0, 0.0};
                    if (m < l) {
#pragma unroll
                        for (int t = 0; t < RPT; ++t) a = c_fmac(c[m][t], c[l][t], a);
                    }
                    y[2 * m] = a.x;
                    y[2 * m + 1] = a.y;
                }
                block_sum_arr<NTP, 2 * PNB>(y, red);
                cd z[PNB];
#pragma unroll
                for (int m = 0; m < PNB; ++m) {
                    z[m] = cd{0.0, 0.0};
                    if (m < l) {
#pragma unroll
                        for (int mm = 0; mm < PNB; ++mm)
                            if (mm <= m) z[m] = c_fmac(Tf[mm][m], cd{y[2 * mm], y[2 * mm + 1]}, z[m]);
                    }
                }
#pragma unroll
                for (int m = 0; m < PNB; ++m)
                    if (m < l) {
                        const cd zn{-z[m].x, -z[m].y};
#pragma unroll
                        for (int t = 0; t < RPT; ++t) c[l][t] = c_fma(c[m][t], zn, c[l][t]);
                    }
            }
            double g[KC];
#pragma unroll
            for (int q = 0; q < KC; ++q) g[q] = 0.0;
#pragma unroll
            for (int t = 0; t < RPT; ++t) {
                const int64_t i = tid + (int64_t)t * NTP;
                if (i > kl && i < M) {
                    g[2 * PNB] = fma(c[l][t].x, c[l][t].x, fma(c[l][t].y, c[l][t].y, g[2 * PNB]));
#pragma unroll
                    for (int m = 0; m < PNB; ++m)
                        if (m < l) {
                            const cd a = c_mulc(c[m][t], c[l][t]);
                            g[2 * m] += a.x;
                            g[2 * m + 1] += a.y;
                        }
                } else if (i == kl) {
                    s_alpha = c[l][t];
#pragma unroll
                    for (int m = 0; m < PNB; ++m)
                        if (m < l) s_vrow[m] = c[m][t];
                }
            }
            block_sum_arr<NTP, KC>(g, red);
            const double s2 = g[2 * PNB];
            const cd alpha = s_alpha;
            double beta = alpha.x;
            cd tk{0.0, 0.0}, scale{0.0, 0.0};
            if (s2 > 0.0 || alpha.y != 0.0) {       // zlarfg
                beta = -copysign(sqrt(alpha.x * alpha.x + alpha.y * alpha.y + s2), alpha.x);
                tk = cd{(beta - alpha.x) / beta, -alpha.y / beta};
                const double dr = alpha.x - beta, di = alpha.y, dn = dr * dr + di * di;
                scale = cd{dr / dn, -di / dn};      // 1 / (alpha - beta)
            }
            // column l of T:  T[i2][l] = -tau sum_{m=i2}^{l-1} T[i2][m] (v_m^H v_l),  v_m^H v_l = g_m scale + conj(v_m[kl])
            if (tid < l) {
                cd acc{0.0, 0.0};
#pragma unroll
                for (int m = 0; m < PNB; ++m)
                    if (m >= tid && m < l) {
                        cd sml = c_mul(cd{g[2 * m], g[2 * m + 1]}, scale);
                        sml.x += s_vrow[m].x;
                        sml.y -= s_vrow[m].y;
                        acc = c_fma(Tf[tid][m], sml, acc);
                    }
                const cd r = c_mul(tk, acc);
                Tf[tid][l] = cd{-r.x, -r.y};
            } else if (tid == l) {
                Tf[l][l] = tk;
                tau[J.c_off + kl] = tk;
            }
            cd *vk = Vall + J.x_off + kl * M;
            cd *xc = Xb + s_p[l] * M;
#pragma unroll
            for (int t = 0; t < RPT; ++t) {
                const int64_t i = tid + (int64_t)t * NTP;
                if (i < M) {
                    cd v;
                    if (i < kl) {
                        v = cd{0.0, 0.0};
                        xc[i] = c[l][t];
                    } else if (i == kl) {
                        xc[i] = cd{beta, 0.0};
                        v = cd{1.0, 0.0};
                    } else {
                        xc[i] = cd{0.0, 0.0};
                        v = c_mul(c[l][t], scale);
                    }
                    vk[i] = v;
                    c[l][t] = v;
                }
            }
            __syncthreads();
        }
    }
    if (tid < PNB * PNB) Tpan[(int64_t)b * PNB * PNB + tid] = Tf[tid / PNB][tid % PNB];
}

template <int NB>
struct WySmemC {
    double redr[NTR / 64][NB][RCOLS], redi[NTR / 64][NB][RCOLS];
    cd Y[NB][RCOLS], Z[NB][RCOLS], T[NB][NB];
    double nrm[NTR / 64][RCOLS];
};

// complex compact-WY tile:  C[k0:, tile] <- (I - V T' V^H) C[k0:, tile],  T' = T^H (CONJT: factorisation) or T (forming Q U_R)
template <int NB, bool CONJT, bool NORMS>
__device__ __forceinline__ double wy_apply_tile_c(cd *Cb, int64_t rs, int64_t cs, int64_t k0, int64_t M, int64_t j0, int64_t jend,
                                                  const cd *__restrict__ V, int nb, int64_t kend, WySmemC<NB> &sm,
                                                  bool col_ok = true) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lo = lane & 15, kq = lane >> 4;
    const int64_t j = j0 + lo;
    const bool jok = (j < jend) && col_ok;
    // ---- pass 1: Y(l, j) = sum_i conj(V(l, i)) C(i, j)
    d4 ar = {0, 0, 0, 0}, ai = {0, 0, 0, 0};
    for (int64_t i0 = k0 + 16 * wave; i0 < M; i0 += 16 * (NTR / 64)) {
        cd a[4], bb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t i = i0 + 4 * kq + q;
            const bool iok = i < M;
            a[q] = (iok && lo < nb) ? V[(int64_t)lo * M + i] : cd{0.0, 0.0};
            bb[q] = (iok && jok) ? Cb[i * rs + j * cs] : cd{0.0, 0.0};
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ar = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q].x, bb[q].x, ar, 0, 0, 0);
            ar = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q].y, bb[q].y, ar, 0, 0, 0);
            ai = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q].x, bb[q].y, ai, 0, 0, 0);
            ai = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[q].y, bb[q].x, ai, 0, 0, 0);
        }
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
        if (kq + 4 * reg < NB) {
            sm.redr[wave][kq + 4 * reg][lo] = ar[reg];
            sm.redi[wave][kq + 4 * reg][lo] = ai[reg];
        }
    __syncthreads();
    if (threadIdx.x < NB * RCOLS) {
        const int l = threadIdx.x >> 4, c = threadIdx.x & 15;
        double tr = 0, ti = 0;
#pragma unroll
        for (int q = 0; q < NTR / 64; ++q) {
            tr += sm.redr[q][l][c];
            ti += sm.redi[q][l][c];
        }
        sm.Y[l][c] = cd{tr, ti};
    }
    __syncthreads();
    if (threadIdx.x < NB * RCOLS) {
        const int l = threadIdx.x >> 4, c = threadIdx.x & 15;
        cd t{0.0, 0.0};
#pragma unroll
        for (int m = 0; m < NB; ++m) t = CONJT ? c_fmac(sm.T[m][l], sm.Y[m][c], t) : c_fma(sm.T[l][m], sm.Y[m][c], t);
        sm.Z[l][c] = cd{-t.x, -t.y};
    }
    __syncthreads();
    cd zneg[NB / 4];
#pragma unroll
    for (int q = 0; q < NB / 4; ++q) zneg[q] = sm.Z[4 * q + kq][lo];
    // ---- pass 2: C += V Zneg
    double nrm = 0;
    for (int64_t i0 = k0 + 16 * wave; i0 < M; i0 += 2 * 16 * (NTR / 64)) {
        d4 cr[2], ci[2];
        cd av[2][NB / 4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t ib = i0 + (int64_t)h * 16 * (NTR / 64);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int64_t i = ib + kq + 4 * reg;
                const cd v = (i < M && jok) ? Cb[i * rs + j * cs] : cd{0.0, 0.0};
                cr[h][reg] = v.x;
                ci[h][reg] = v.y;
            }
#pragma unroll
            for (int q = 0; q < NB / 4; ++q) {
                const int64_t i = ib + lo;
                const int l = 4 * q + kq;
                av[h][q] = (i < M && l < nb) ? V[(int64_t)l * M + i] : cd{0.0, 0.0};
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < NB / 4; ++q) {
                cr[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[h][q].x, zneg[q].x, cr[h], 0, 0, 0);
                cr[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[h][q].y, zneg[q].y, cr[h], 0, 0, 0);
                ci[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[h][q].x, zneg[q].y, ci[h], 0, 0, 0);
                ci[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[h][q].y, zneg[q].x, ci[h], 0, 0, 0);
            }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t ib = i0 + (int64_t)h * 16 * (NTR / 64);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int64_t i = ib + kq + 4 * reg;
                if (i < M && jok) {
                    Cb[i * rs + j * cs] = cd{cr[h][reg], ci[h][reg]};
                    if (NORMS && i >= kend) nrm = fma(cr[h][reg], cr[h][reg], fma(ci[h][reg], ci[h][reg], nrm));
                }
            }
        }
    }
    if (!NORMS) return 0.0;
    nrm += __shfl_xor(nrm, 16, 64);
    nrm += __shfl_xor(nrm, 32, 64);
    if (lane < RCOLS) sm.nrm[wave][lo] = nrm;
    __syncthreads();
    double t = 0;
    if (threadIdx.x < RCOLS) {
#pragma unroll
        for (int q = 0; q < NTR / 64; ++q) t += sm.nrm[q][threadIdx.x];
    }
    return t;
}

__global__ __launch_bounds__(NTR) void qrp_update_kernel_c(const QrpJob *__restrict__ jobs, int k, cd *__restrict__ X,
                                                           const cd *__restrict__ Vall, double *__restrict__ cn,
                                                           const QrpState *__restrict__ state, const cd *__restrict__ Tpan) {
    __shared__ WySmemC<PNB> sm;
    const int b = blockIdx.y;
    const QrpState st = state[b];
    if (st.done || st.nbk == 0) return;
    const QrpJob J = jobs[b];
    const int nbk = st.nbk;
    const int64_t j0 = (int64_t)blockIdx.x * RCOLS;
    if (j0 >= J.N) return;
    const int64_t jl = j0 + (threadIdx.x & (RCOLS - 1));
    const bool col_ok = (jl < J.N) && (cn[J.c_off + jl] >= 0.0);
    if (__ballot(col_ok) == 0) return;
    if (threadIdx.x < PNB * PNB) sm.T[threadIdx.x / PNB][threadIdx.x % PNB] = Tpan[(int64_t)b * PNB * PNB + threadIdx.x];
    const double nrm = wy_apply_tile_c<PNB, true, true>(X + J.x_off, 1, J.M, k, J.M, j0, J.N, Vall + J.x_off + (int64_t)k * J.M,
                                                        nbk, (int64_t)k + nbk, sm, col_ok);
    if (threadIdx.x < RCOLS && col_ok) cn[J.c_off + jl] = nrm;
}

__global__ __launch_bounds__(NT) void qrp_extract_kernel_c(const QrpJob *__restrict__ jobs, const QrpState *__restrict__ state,
                                                           const cd *__restrict__ X, const int64_t *__restrict__ cperm,
                                                           cd *__restrict__ Rtop) {
    const QrpJob J = jobs[blockIdx.y];
    const int64_t r = state[blockIdx.y].rank, N = J.N;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < r * N; e += (int64_t)gridDim.x * NT) {
        const int64_t i = e / N, j = e - i * N;
        Rtop[J.r_off + e] = (j >= i) ? X[J.x_off + cperm[J.c_off + j] * J.M + i] : cd{0.0, 0.0};
    }
}

__global__ __launch_bounds__(NT) void qrp_form_t_kernel_c(const QrpJob *__restrict__ jobs, const QrpState *__restrict__ state,
                                                          const cd *__restrict__ UR, cd *__restrict__ T) {
    const QrpJob J = jobs[blockIdx.y];
    const int64_t r = state[blockIdx.y].rank, M = J.M;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < M * r; e += (int64_t)gridDim.x * NT) {
        const int64_t i = e / r;
        T[J.x_off + e] = (i < r) ? UR[J.r_off + e] : cd{0.0, 0.0};
    }
}

__global__ __launch_bounds__(NTR) void qrp_tfactor_kernel_c(const QrpJob *__restrict__ jobs, const QrpState *__restrict__ state,
                                                            const cd *__restrict__ Vall, const cd *__restrict__ tau,
                                                            cd *__restrict__ Tfac) {
    __shared__ cd S[QNB][QNB + 1];
    __shared__ cd Tf[QNB][QNB + 1];
    const QrpJob J = jobs[blockIdx.y];
    const int64_t r = state[blockIdx.y].rank, M = J.M;
    const int64_t k0 = (int64_t)blockIdx.x * QNB;
    if (k0 >= r) return;
    const int nb = (int)((r - k0 < QNB) ? (r - k0) : QNB);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const cd *V = Vall + J.x_off + k0 * M;
    for (int q = 0; q < QNB; ++q) {          // S[l][l'] = v_l^H v_l'
        const int l = wave, lp = q;
        cd acc{0.0, 0.0};
        if (l < lp && lp < nb)
            for (int64_t i = k0 + lp + lane; i < M; i += 64) acc = c_fmac(V[l * M + i], V[lp * M + i], acc);
        acc.x = wave_sum(acc.x);
        acc.y = wave_sum(acc.y);
        if (lane == 0) S[l][lp] = acc;
    }
    __syncthreads();
    if (threadIdx.x < QNB * QNB) Tf[threadIdx.x >> 4][threadIdx.x & 15] = cd{0.0, 0.0};
    __syncthreads();
    for (int j = 0; j < nb; ++j) {
        const cd tj = tau[J.c_off + k0 + j];
        if (threadIdx.x < j) {
            const int i = threadIdx.x;
            cd acc{0.0, 0.0};
            for (int m = i; m < j; ++m) acc = c_fma(Tf[i][m], S[m][j], acc);
            const cd rr = c_mul(tj, acc);
            Tf[i][j] = cd{-rr.x, -rr.y};
        } else if (threadIdx.x == j)
            Tf[j][j] = tj;
        __syncthreads();
    }
    if (threadIdx.x < QNB * QNB)
        Tfac[(J.pad0 + blockIdx.x) * (QNB * QNB) + threadIdx.x] = Tf[threadIdx.x >> 4][threadIdx.x & 15];
}

__global__ __launch_bounds__(NTR) void qrp_apply_q_block_kernel_c(const QrpJob *__restrict__ jobs,
                                                                  const QrpState *__restrict__ state, int blk,
                                                                  cd *__restrict__ C, const cd *__restrict__ Vall,
                                                                  const cd *__restrict__ Tfac) {
    __shared__ WySmemC<QNB> sm;
    const int b = blockIdx.y;
    const QrpJob J = jobs[b];
    const int64_t r = state[b].rank;
    const int64_t k0 = (int64_t)blk * QNB;
    if (k0 >= r) return;
    const int64_t j0 = (int64_t)blockIdx.x * RCOLS;
    if (j0 >= r) return;
    const int nb = (int)((r - k0 < QNB) ? (r - k0) : QNB);
    if (threadIdx.x < QNB * QNB) sm.T[threadIdx.x >> 4][threadIdx.x & 15] = Tfac[(J.pad0 + blk) * (QNB * QNB) + threadIdx.x];
    wy_apply_tile_c<QNB, false, false>(C + J.x_off, r, 1, k0, J.M, j0, r, Vall + J.x_off + k0 * J.M, nb, 0, sm);
}

__global__ __launch_bounds__(NT) void qrp_output_kernel_c(const QrpJob *__restrict__ jobs, const SvdJob *__restrict__ sj,
                                                          const QrpState *__restrict__ state, const cd *__restrict__ T,
                                                          const double *__restrict__ SR, const cd *__restrict__ VHR,
                                                          const int64_t *__restrict__ cperm, cd *__restrict__ U,
                                                          double *__restrict__ S, cd *__restrict__ VH) {
    const QrpJob J = jobs[blockIdx.y];
    const SvdJob O = sj[blockIdx.y];
    const int64_t r = state[blockIdx.y].rank, M = J.M, N = J.N, K = N;
    const int64_t stride = (int64_t)gridDim.x * NT, t0 = (int64_t)blockIdx.x * NT + threadIdx.x;
    const cd zero{0.0, 0.0};
    for (int64_t e = t0; e < K; e += stride) S[O.s_off + e] = (e < r) ? SR[J.c_off + e] : 0.0;
    if (!J.tr) {
        for (int64_t e = t0; e < M * K; e += stride) {
            const int64_t i = e / K, jj = e - i * K;
            U[O.u_off + e] = (jj < r) ? T[J.x_off + i * r + jj] : zero;
        }
        for (int64_t e = t0; e < K * N; e += stride) {
            const int64_t jj = e / N, c = e - jj * N;
            VH[O.vh_off + jj * N + cperm[J.c_off + c]] = (jj < r) ? VHR[J.r_off + jj * N + c] : zero;
        }
    } else {
        for (int64_t e = t0; e < N * K; e += stride) {
            const int64_t c = e / K, jj = e - c * K;
            cd v = (jj < r) ? VHR[J.r_off + jj * N + c] : zero;
            v.y = -v.y;
            U[O.u_off + cperm[J.c_off + c] * K + jj] = v;
        }
        for (int64_t e = t0; e < K * M; e += stride) {
            const int64_t jj = e / M, c = e - jj * M;
            cd v = (jj < r) ? T[J.x_off + c * r + jj] : zero;
            v.y = -v.y;
            VH[O.vh_off + e] = v;
        }
    }
}

int tpa_svd_use_qrp = 1;   // real data: rank-revealing pivoted QR before the Jacobi iteration

int tpa_svd_small_panel = 1;   // test hook (TPA_SVD_SMALL_PANEL=0): one-wavefront panel kernel for blocks <= 512 x 512
int tpa_svd_wide_round = 1;    // real data, every block <= 2048 columns of [W | G]: one 1024-thread workgroup per pair (svd_round_wide_kernel)
int tpa_svd_fused_round = 1;   // real data: one launch per Jacobi round (sibling workgroups synchronise through a counter)
int tpa_svd_local_sweeps = 1;
int tpa_svd_predict_convergence = 1;   // a sweep without "big" rotations (scaled cosine > 1e-7) ends the iteration: no verification sweep (validated on the MI355X in round 2)
int tpa_svd_cross_only = 1;  // rounds r > 0 of a sweep rotate only cross-block pairs
                        // 32 x 32 solve dominates); kept as a tuning option, off by default
int tpa_svd_force_pairwise = 0;  // test hook: 1 = use the wavefront-per-pair kernel also for real data
int tpa_svd_rank_cap = 0;        // > 0: the pivoted QR gives up (TPA_E_RANKCAP) once a block needs more than this many columns
int tpa_svd_lookahead = 1;       // 32-row-block path: first round of the next sweep enqueued before the host reads the counters of this one (bit 13 switches it off)
int tpa_svd_b32 = 1;             // real data: 32-row blocks, three launches per round (tpa_svd_b32.inc); bit 12 of tpa_svd_set_algorithm switches it off
int tpa_svd_gonly = 1;           // real data, 32-row blocks: Gram-only sweeps (one exact Gram matrix per sweep, rounds on it alone, one product
                                 // [W | G] <- Qtot [W | G] at the end; tpa_svd_b32.inc); bit 20 of tpa_svd_set_algorithm switches it off
int tpa_svd_overlap_c = 0;       // complex Gram-only rounds: tiles that the next solve does not read on a second stream (bit 14; measured slower)
constexpr int64_t REF_MIN_R = 96;   // calls whose largest block has fewer rows keep the plain Jacobi rounds (1 - 2 rounds per sweep there)

int tpa_svd_fused_rounds = 1;    // Gram-only sweeps: one launch per round (svd_b32_round_kernel); bit 23 of tpa_svd_set_algorithm: solve and update as two launches
inline void b32_solve_launch(int n_pairs, hipStream_t st, const SvdJob *jobs, const B32Pair *b32p, const double *b32g, double *b32q, int *b32f,
                             unsigned int *cnt, const double *fro2, double rho, int full_local, const double *sfull, int r) {
    svd_b32_solve2_kernel<<<n_pairs, NTS3, 0, st>>>(jobs, b32p, b32g, b32q, b32f, cnt, fro2, rho, full_local, sfull, r);
}

// ---- round 6: the rounds of a Gram-only sweep as TABLES built from the activity of the block pairs ------------------------------------
// eigh = shift + SVD: after the shift every singular value sits within |A| of mu -- a (nearly) degenerate cluster, where the iteration does
// NOT converge quadratically and a sweep without big rotations still leaves cosines of ~1e-8 (tests/test_kernels_gpu.py::
// test_eigh_batch_mixer_blocks: 7.6e-9 at 570 / 1086 rows).  tpa_eigh_batch therefore switches the predicted-convergence exit off for its
// call: the iteration ends only when the exact Gram matrix of a sweep start shows no pair left to rotate.
static thread_local int tpa_svd_strict = 0;
// eigh as a TWO-SIDED block Jacobi iteration of its own (round 6, tpa_eigh_batch): the rounds of a Gram-only sweep ARE a two-sided Jacobi
// method on a Hermitian matrix S (solve: cyclic Jacobi on the 64 x 64 diagonal block of a pair; update: S[P, P'] <- Q_P S[P, P'] Q_P'^H,
// Qtot[P, :] <- Q_P Qtot[P, :]).  For a Hermitian input the SVD detour -- S = W W^H by a GEMM at the start of every sweep, [W | G] <- Qtot
// [W | G] by two more at its end -- buys nothing: with `tpa_svd_direct` S starts as the (shifted) matrix itself, every sweep continues on the
// S the previous one left, Qtot accumulates over the whole call, eigenvalues = diag(S), eigenvectors = rows of Qtot.  No GEMM at all
// (a third of the flops of a sweep), no squaring of the spectrum.  The stopping rule |S_ij| <= eps sqrt(n) sqrt(S_ii S_jj) reads
// |H_ij| <= eps sqrt(n) mu on the shifted matrix: the absolute accuracy class of LAPACK's eigh, as before.
static thread_local int tpa_svd_direct = 0;        // request (set around svd_run by tpa_eigh_batch)
static thread_local int tpa_svd_direct_used = 0;   // answer: the call ran the direct iteration (eigenvectors come out on the G side)
int tpa_eigh_direct = 1;                           // test hook (TPA_EIGH_DIRECT=0 / tpa_eigh_set_direct): the shift + one-sided SVD route of rounds 1 - 5
int tpa_svd_dyn_round0 = 1;      // the first round of a sweep adapts to the activity of its pairs (bit 25 of tpa_svd_set_algorithm: off)
int tpa_svd_dyn = 1;             // 0 (TPA_SVD_DYN=0 / bit 24 of tpa_svd_set_algorithm): the full round-robin schedule in every sweep (rounds 3 - 5)
int tpa_svd_apply_skip = getenv("TPA_SVD_APPLY_SKIP") ? atoi(getenv("TPA_SVD_APPLY_SKIP")) : 1;      // activity-driven sweeps: tiles of Qtot [W | G] whose rows did not rotate are copied (tpa_gemm.hip: identity-row skip)
int64_t tpa_svd_dyn_rounds = 0, tpa_svd_dyn_rounds_static = 0, tpa_svd_dyn_sweeps = 0;      // statistics (tpa_svd_dyn_stats)

inline void b32_host_pair_of(int R, int pair, int round, int &bi, int &bj) {      // = b32_pair_of_R on the device
    const int NB = (R + BB - 1) / BB, NBp = (NB + 1) / 2 * 2, mod = NBp - 1;
    const int r = (mod > 0) ? (round % mod) : 0;
    if (pair == 0) {
        bi = NBp - 1;
        bj = r;
    } else {
        bi = (r + pair) % mod;
        bj = (r - pair + mod) % mod;
    }
    if (bi > bj) std::swap(bi, bj);
}

typedef std::vector<std::pair<int, int>> B32Matching;      // NBp / 2 pairs (bi < bj): a perfect matching of the padded blocks of one job

// The rounds 1, 2, ... of one job for this sweep (round 0 is always the static first round: it rotates ALL row pairs inside its block
// pairs).  `act`: NBp x NBp words of b32_activity_kernel (upper triangle).  Every round is a perfect matching that contains as many of
// the still unscheduled active block pairs as a greedy pass finds (vertices by remaining degree); the other blocks are paired among
// themselves (their solves find nothing to rotate and leave at once).  Nearly complete activity: the round-robin schedule itself.
inline void b32_job_rounds(int R, const int *act, int max_rounds, std::vector<B32Matching> &out) {
    const int NB = (R + BB - 1) / BB, NBp = (NB + 1) / 2 * 2, np = NBp / 2;
    out.clear();
    if (NBp < 4) return;         // one pair: round 0 is everything
    std::vector<char> A((size_t)NBp * NBp, 0);
    int n_edges = 0;
    for (int i = 0; i < NB; ++i)
        for (int j = i + 1; j < NB; ++j)
            if (act[i * NBp + j]) {
                A[i * NBp + j] = A[j * NBp + i] = 1;
                ++n_edges;
            }
    for (int p = 0; p < np; ++p) {      // round 0 covers its own pairs
        int bi, bj;
        b32_host_pair_of(R, p, 0, bi, bj);
        if (bj < NB && A[bi * NBp + bj]) {
            A[bi * NBp + bj] = A[bj * NBp + bi] = 0;
            --n_edges;
        }
    }
    if (n_edges == 0) return;
    if (10 * n_edges > 6 * (NB * (NB - 1) / 2)) {      // dense: the round-robin rounds 1 .. NBp - 2 (a greedy colouring would need more)
        for (int r = 1; r < NBp - 1 && (int)out.size() < max_rounds; ++r) {
            B32Matching m(np);
            for (int p = 0; p < np; ++p) b32_host_pair_of(R, p, r, m[p].first, m[p].second);
            out.push_back(m);
        }
        return;
    }
    std::vector<int> deg(NBp), order(NBp);
    std::vector<char> used(NBp);
    while (n_edges > 0 && (int)out.size() < max_rounds) {
        for (int v = 0; v < NBp; ++v) {
            deg[v] = 0;
            for (int w = 0; w < NBp; ++w) deg[v] += A[v * NBp + w];
            order[v] = v;
            used[v] = 0;
        }
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return deg[x] > deg[y]; });
        B32Matching m;
        for (int v : order) {
            if (used[v] || deg[v] == 0) continue;
            int best = -1;
            for (int w = 0; w < NBp; ++w)
                if (!used[w] && w != v && A[v * NBp + w] && (best < 0 || deg[w] > deg[best])) best = w;
            if (best < 0) continue;
            m.push_back({std::min(v, best), std::max(v, best)});
            used[v] = used[best] = 1;
            A[v * NBp + best] = A[best * NBp + v] = 0;
            --n_edges;
        }
        int prev = -1;
        for (int v = 0; v < NBp; ++v) {      // the blocks without a partner this round: paired in index order
            if (used[v]) continue;
            if (prev < 0)
                prev = v;
            else {
                m.push_back({prev, v});
                prev = -1;
            }
        }
        out.push_back(m);
    }
}

struct Layout {
    std::vector<SvdJob> jobs;
    std::vector<int2> rows;   // (job,row) per wavefront, padded to multiple of 4 with (-1,-1)
    std::vector<int2> pairs;  // (job,pair)
    std::vector<BEntry> bentries;  // (job, pair, part, nparts) for the split rounds
    std::vector<int2> wpairs;      // (job, block pair) for the one-workgroup-per-pair round
    std::vector<B32Entry> b32_entries;   // (job, pair, part) of the 32-row-block rounds
    std::vector<B32Pair> b32_pairs;
    int64_t nb32_max_pad = 0;
    int64_t off_b32e = 0, off_b32p = 0, off_b32g = 0, off_b32q = 0, off_b32f = 0, off_tab_end = 0;
    std::vector<B32GUp> b32_gup;   // tiles of the Gram-only rounds
    int n_gup_s = 0;
    std::vector<int> b32_first_pair;
    int64_t off_gup = 0;
    std::vector<B32Act> b32_act;         // round 6: block pairs (bi <= bj) whose activity the exact Gram matrix of a sweep start decides
    std::vector<int> b32_act_off;        // per job: offset of its NBp x NBp activity words
    int64_t n_act = 0, off_act_ents = 0, off_act = 0, off_sched = 0;
    int64_t n_rot = 0, off_rot = 0;      // words "this 64-row tile rotated in the sweep" (b32_rot_word): real data, activity-driven rounds
    RefTables ref;                 // GEMM tables of the Gram-only sweeps (empty unless the largest block has >= REF_MIN_R rows)
    int64_t off_rtasks = 0, off_rlinks = 0, off_rtiles = 0, off_rrt = 0;                // ... inside the uploaded table range
    int64_t off_w2 = 0, off_rp = 0, off_rq = 0, off_rm = 0;           // second [W | G] image, split-K partials, Qtot, S
    bool wide_ok = true;           // every job has <= FIT * NTW / 64 column chunks of [W | G]
    int64_t nb_max_pad = 0;
    int64_t max_part_chunks = 0;   // largest number of (W + G) column chunks of one part (fused round: <= 4 * FIT)
    int64_t w_elems = 0, g_elems = 0, sig_elems = 0, rmax_pad = 0;
    // byte offsets inside work buffer
    int64_t off_w = 0, off_g = 0, off_sig = 0, off_perm = 0, off_jobs = 0, off_rows = 0, off_pairs = 0,
            off_bent = 0, off_wpairs = 0, off_pcnt = 0, off_gpart = 0, off_cnt = 0, off_fro = 0, off_fpart = 0, total = 0;
};

inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

Layout build_layout(int dtype, const int64_t *jobs_host, int n_jobs) {
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
        J.tr = (J.m > J.n || (J.m == J.n && !(j[6] & 1))) ? 1 : 0;
        J.fro_bits = j[7];
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
        {
            const int64_t NB = (J.R + 7) / 8, NBp = (NB + 1) / 2 * 2;
            // column parts: >= 4 chunks of 64 columns each, at most 8 parts
            const int64_t nchunk = (J.L + 63) / 64;   // (complex kernels use 32-column chunks: twice as many)
            // real kernels: W and G chunks of a part are dealt out to 4 wavefronts together -> aim at <= 4 per part
            const int64_t nch_all = nchunk + (J.R + 63) / 64;
            int nparts = (dtype == TPA_C128) ? (int)std::min<int64_t>(8, std::max<int64_t>(1, nchunk / 4))
                                             : (int)std::min<int64_t>(8, std::max<int64_t>(1, (nch_all + 7) / 8));
            if (dtype != TPA_C128) {   // every part must fit the LDS-resident budget of the fused round (4 * FIT chunks)
                const int64_t nchG0 = (J.R + 63) / 64;
                auto worst = [&](int np) {
                    int64_t w = 0;
                    for (int q = 0; q < np; ++q)
                        w = std::max(w, (nchunk * (q + 1) / np - nchunk * q / np) + (nchG0 * (q + 1) / np - nchG0 * q / np));
                    return w;
                };
                while (nparts < 8 && worst(nparts) > 8) ++nparts;
            } else {                   // complex fused round: chunks of 32 columns, <= 4 * FITC register-resident chunks per part
                const int64_t nW0 = (J.L + CHC - 1) / CHC, nG0 = (J.R + CHC - 1) / CHC;
                auto worst = [&](int np) {
                    int64_t w = 0;
                    for (int q = 0; q < np; ++q) w = std::max(w, (nW0 * (q + 1) / np - nW0 * q / np) + (nG0 * (q + 1) / np - nG0 * q / np));
                    return w;
                };
                nparts = (int)std::min<int64_t>(8, std::max<int64_t>(1, (nW0 + nG0 + 4 * FITC - 1) / (4 * FITC)));
                while (nparts < 8 && worst(nparts) > 4 * FITC) ++nparts;
                if (NBp >= 2) lay.max_part_chunks = std::max(lay.max_part_chunks, worst(nparts));
            }
            for (int64_t p = 0; p < NBp / 2; ++p)
                for (int q = 0; q < nparts; ++q) lay.bentries.push_back(BEntry{b, (int)p, q, nparts});
            for (int64_t p = 0; p < NBp / 2; ++p) lay.wpairs.push_back(int2{b, (int)p});
            {   // 32-row blocks
                const int64_t NB32 = (J.R + BB - 1) / BB, NB32p = (NB32 + 1) / 2 * 2;
                const int64_t nW32 = (J.L + CB - 1) / CB, nG32 = (J.R + CB - 1) / CB;
                const int np32 = (int)std::min<int64_t>(B32_MAX_PARTS, std::max<int64_t>(1, (nW32 + nG32 + 3) / 4));
                lay.b32_first_pair.push_back((int)lay.b32_pairs.size());
                for (int64_t p = 0; p < NB32p / 2; ++p) {
                    const int pairidx = (int)lay.b32_pairs.size();
                    lay.b32_pairs.push_back(B32Pair{b, (int)p, (int)lay.b32_entries.size(), np32, (int)J.R, (int)J.L, J.g_off});
                    for (int q = 0; q < np32; ++q) lay.b32_entries.push_back(B32Entry{b, (int)p, q, np32, pairidx, 0, 0, 0});
                }
                lay.nb32_max_pad = std::max(lay.nb32_max_pad, NB32p);
            }
            if (nch_all > FIT * (NTW / 64)) lay.wide_ok = false;
            if (dtype != TPA_C128 && NBp >= 2) {
                const int64_t nchG = (J.R + 63) / 64;
                for (int q = 0; q < nparts; ++q) {
                    const int64_t nw = nchunk * (q + 1) / nparts - nchunk * q / nparts;
                    const int64_t ng = nchG * (q + 1) / nparts - nchG * q / nparts;
                    lay.max_part_chunks = std::max(lay.max_part_chunks, nw + ng);
                }
            }
            lay.nb_max_pad = std::max(lay.nb_max_pad, NBp);
        }
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
    lay.off_bent = o;
    o = align_up(o + (int64_t)lay.bentries.size() * sizeof(BEntry), 256);
    lay.off_wpairs = o;
    o = align_up(o + (int64_t)lay.wpairs.size() * sizeof(int2), 256);
    lay.off_b32e = o;            // (the host-built tables off_jobs .. off_tab_end are contiguous: ONE staged upload, svd_upload_tables)
    o = align_up(o + (int64_t)lay.b32_entries.size() * sizeof(B32Entry), 256);
    lay.off_b32p = o;
    o = align_up(o + (int64_t)lay.b32_pairs.size() * sizeof(B32Pair), 256);
    if (lay.rmax_pad >= REF_MIN_R) {     // (independent of the algorithm switches: tpa_svd_worksize must not depend on them)
        ref_make_tables(lay.ref, lay.jobs, dtype, (lay.g_elems + 1) / 2 * 2, (lay.off_g - lay.off_w) / esz);
        lay.off_rtasks = o;
        o = align_up(o + (int64_t)lay.ref.tasks.size() * 8, 256);
        lay.off_rlinks = o;
        o = align_up(o + (int64_t)lay.ref.links.size() * 8, 256);
        lay.off_rtiles = o;
        o = align_up(o + (int64_t)lay.ref.tiles.size() * 4, 256);
        lay.off_rrt = o;
        o = align_up(o + (int64_t)lay.ref.rtiles.size() * sizeof(RefTile), 256);
        {
            for (int b = 0; b < (int)lay.jobs.size(); ++b) {
                const SvdJob &J = lay.jobs[b];
                const int NB32 = (int)((J.R + BB - 1) / BB), np = (NB32 + 1) / 2, first = lay.b32_first_pair[b];
                for (int pa = 0; pa < np; ++pa)      // (complex: the Hermitian mirror tile is written by the tile's own workgroup)
                    for (int pb = (dtype == TPA_C128) ? pa : 0; pb < np; ++pb) lay.b32_gup.push_back(B32GUp{2 * b, pa, pb, first + pa, first + pb, (int)J.R, J.g_off});
            }
            lay.n_gup_s = (int)lay.b32_gup.size();       // S tiles first, then the Qtot tiles (the last launch of a sweep needs only those)
            for (int b = 0; b < (int)lay.jobs.size(); ++b) {
                const SvdJob &J = lay.jobs[b];
                const int NB32 = (int)((J.R + BB - 1) / BB), np = (NB32 + 1) / 2, first = lay.b32_first_pair[b];
                const int nct = (int)((J.R + TB - 1) / TB);
                for (int pa = 0; pa < np; ++pa)
                    for (int c = 0; c < nct; ++c) lay.b32_gup.push_back(B32GUp{2 * b + 1, pa, c, first + pa, first + pa, (int)J.R, J.g_off});
            }
            lay.off_gup = o;
            o = align_up(o + (int64_t)lay.b32_gup.size() * sizeof(B32GUp), 256);
            if (dtype != TPA_C128) {      // activity entries of the dynamic schedule (real data)
                for (int b = 0; b < (int)lay.jobs.size(); ++b) {
                    const SvdJob &J = lay.jobs[b];
                    const int NB32 = (int)((J.R + BB - 1) / BB), NBp = (NB32 + 1) / 2 * 2;
                    lay.b32_act_off.push_back((int)lay.n_act);
                    for (int bi = 0; bi < NB32; ++bi)
                        for (int bj = bi; bj < NB32; ++bj) lay.b32_act.push_back(B32Act{b, bi, bj, (int)lay.n_act + bi * NBp + bj});
                    lay.n_act += (int64_t)NBp * NBp;
                }
                lay.off_act_ents = o;
                o = align_up(o + (int64_t)lay.b32_act.size() * sizeof(B32Act), 256);
            }
        }
    }
    lay.off_tab_end = o;
    lay.off_pcnt = o;
    o = align_up(o + (int64_t)lay.bentries.size() * 4 + 64, 256);   // per-entry pair counters + error flag
    lay.off_gpart = o;
    o = align_up(o + (int64_t)lay.bentries.size() * 512 * 8, 256);   // 256 (real) / 512 (complex: re + im) doubles per entry
    lay.off_cnt = o;
    o = align_up(o + 256, 256);
    lay.off_fro = o;
    o = align_up(o + (int64_t)lay.jobs.size() * 8, 256);
    lay.off_fpart = o;
    o = align_up(o + (int64_t)lay.jobs.size() * 64 * 8, 256);
    lay.off_b32f = o;
    o = align_up(o + 2 * (int64_t)lay.b32_pairs.size() * 4, 256);      // (two images: the fused rounds read round k - 1 while they write round k)
    if (dtype != TPA_C128) {
        lay.off_b32g = o;
        o = align_up(o + (int64_t)lay.b32_entries.size() * GSZ * 8, 256);
    }
    lay.off_b32q = o;            // two images of 64 x 64 transforms per pair
    o = align_up(o + ((dtype == TPA_C128) ? 4 : 2) * (int64_t)lay.b32_pairs.size() * TB * TB * 8, 256);
    if (lay.ref.enabled && !lay.b32_act.empty()) {
        for (const SvdJob &J : lay.jobs) lay.n_rot = std::max<int64_t>(lay.n_rot, J.g_off / ROT_GRAIN + (J.R + 63) / 64 + 1);
        lay.off_rot = o;
        o = align_up(o + lay.n_rot * 4, 256);
        lay.off_act = o;
        o = align_up(o + lay.n_act * 4, 256);
        lay.off_sched = o;               // (nb32_max_pad + 2) rounds of one B32Sched per pair
        o = align_up(o + (lay.nb32_max_pad + 2) * (int64_t)lay.b32_pairs.size() * sizeof(B32Sched), 256);
    }
    if (lay.ref.enabled) {
        lay.off_w2 = o;                                   // [W2 | G2] with the spacing of [W | G]
        o = align_up(o + (lay.off_g - lay.off_w) + lay.g_elems * esz, 256);
        lay.off_rp = o;
        o = align_up(o + (int64_t)REF_MAX_SPLIT * ((lay.g_elems + 1) / 2 * 2) * esz, 256);
        lay.off_rq = o;
        o = align_up(o + lay.g_elems * esz, 256);
        lay.off_rm = o;
        o = align_up(o + lay.g_elems * esz, 256);
    }
    lay.total = o;
    return lay;
}

// The layout of a call depends on (dtype, job table) alone -- not on the algorithm switches, see above -- and building it costs 30 - 70 us of
// host time (one table entry per row, row pair, 8- and 32-row block pair, Gram / apply GEMM tile) with the device idle: every npc.svd of a
// steady-state sweep built the layouts of the SAME ~200 job tables again, twice (tpa_svd_worksize, then tpa_svd_batch).  Memoised per host
// thread, least recently used entry out first (profiles/r06_idle_gap_analysis.txt: 190 us between the end of the Lanczos run and the
// first upload of the SVD section).
template <class T>
struct LayoutCache {
    struct Ent {
        std::shared_ptr<const T> lay;
        uint64_t stamp;
    };
    std::unordered_map<std::string, Ent> map;
    uint64_t clock = 0;
    static constexpr size_t CAP = 1024;      // x 0.1 - 0.5 MB per layout of a chi = 2048 call
    template <class F>
    std::shared_ptr<const T> get(int dtype, const int64_t *jobs_host, int n_jobs, F build) {
        std::string key((size_t)n_jobs * 64 + 1, '\0');
        key[0] = (char)dtype;
        std::memcpy(&key[1], jobs_host, (size_t)n_jobs * 64);
        auto it = map.find(key);
        if (it != map.end()) {
            it->second.stamp = ++clock;
            return it->second.lay;
        }
        if (map.size() >= CAP) {      // evict the older half (rare: a sweep touches ~200 - 400 job tables)
            std::vector<uint64_t> stamps;
            stamps.reserve(map.size());
            for (auto &kv : map) stamps.push_back(kv.second.stamp);
            std::nth_element(stamps.begin(), stamps.begin() + stamps.size() / 2, stamps.end());
            const uint64_t cut = stamps[stamps.size() / 2];
            for (auto q = map.begin(); q != map.end();) q = (q->second.stamp < cut) ? map.erase(q) : std::next(q);
        }
        std::shared_ptr<const T> lay = std::make_shared<const T>(build());
        map.emplace(std::move(key), Ent{lay, ++clock});
        return lay;
    }
};
static int tpa_layout_cache_on = getenv("TPA_SVD_LAYOUT_CACHE") ? atoi(getenv("TPA_SVD_LAYOUT_CACHE")) : 1;      // test hook: 0 = rebuild every time

std::shared_ptr<const Layout> make_layout(int dtype, const int64_t *jobs_host, int n_jobs) {
    auto build = [&]() {
        Layout l = build_layout(dtype, jobs_host, n_jobs);
        if (tpa_layout_cache_on) {      // the entry stays: give the growth slack of the push_backs back
            l.rows.shrink_to_fit(), l.pairs.shrink_to_fit(), l.bentries.shrink_to_fit(), l.wpairs.shrink_to_fit();
            l.b32_entries.shrink_to_fit(), l.b32_pairs.shrink_to_fit(), l.b32_gup.shrink_to_fit(), l.b32_act.shrink_to_fit();
            l.ref.tasks.shrink_to_fit(), l.ref.links.shrink_to_fit(), l.ref.tiles.shrink_to_fit(), l.ref.rtiles.shrink_to_fit();
        }
        return l;
    };
    if (!tpa_layout_cache_on) return std::make_shared<const Layout>(build());
    static thread_local LayoutCache<Layout> cache;
    return cache.get(dtype, jobs_host, n_jobs, build);
}

// number of svd_round_fused_kernel workgroups that are guaranteed to be resident together (occupancy query x CUs)
int64_t fused_round_capacity() {
    static int64_t cap = -1;
    if (cap < 0) {
        int per_cu = 0, dev_id = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev_id) != hipSuccess || hipGetDeviceProperties(&prop, dev_id) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, svd_round_fused_kernel, NTG, 0) != hipSuccess)
            cap = 0;
        else
            cap = (int64_t)per_cu * prop.multiProcessorCount;
    }
    return cap;
}

int64_t fused_round_capacity_c() {
    static int64_t cap = -1;
    if (cap < 0) {
        int per_cu = 0, dev_id = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev_id) != hipSuccess || hipGetDeviceProperties(&prop, dev_id) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, svd_round_fused_kernel_c, NTG, 0) != hipSuccess)
            cap = 0;
        else
            cap = (int64_t)per_cu * prop.multiProcessorCount;
    }
    return cap;
}

// ---- pinned staging of the host-built tables and of the few words the host reads back ----------------------------------------
// hipMemcpyAsync from / to PAGEABLE memory is a blocking host call (the runtime stages the data and waits), and the block SVD
// issued 8 - 10 of them per call in front of its launch chains: the round-3 idle-gap analysis (scripts/gap_analysis.py) shows
// ~18 of them per bond update with the GPU idle for ~45 us each.  Now every upload goes through ONE pinned arena per host
// thread (bump allocation, reset at the entry of tpa_svd_batch / tpa_eigh_batch, whose previous call ended with a stream
// synchronisation), all tables of a run in ONE copy, and the convergence counters are posted by a 1-thread kernel into mapped
// pinned memory.
struct PinStage {
    char *base = nullptr;
    size_t cap = 0, used = 0;
    std::vector<char *> retired;      // outgrown arenas: pointers handed out earlier in the same call stay valid until reset()
    void reset() {
        used = 0;
        if (!retired.empty()) {
            (void)hipDeviceSynchronize();     // rare (the arena grew during the previous call); a call that failed half-way may have left copies queued
            for (char *p : retired) (void)hipHostFree(p);
            retired.clear();
        }
    }
    char *take(size_t bytes, hipStream_t st) {
        (void)st;
        const size_t need = (used + 255) / 256 * 256;
        if (base == nullptr || need + bytes > cap) {
            // never free an arena inside a call -- tables staged earlier may still be queued for upload, and the host
            // sort of the singular values reads a block taken before the one that triggered the growth
            if (base != nullptr) retired.push_back(base);
            base = nullptr;
            // (32 MB floor: a hipHostMalloc of a few MB costs ~80 ms on the MI355X box -- measured in round 5 as the one-off 80 - 90 ms
            //  calls of scripts/svd_sketch_bench.py -- so the arena must not grow in 4 MB steps during the first calls of a process)
            cap = std::max<size_t>(2 * (need + bytes), (size_t)32 << 20);
            if (hipHostMalloc((void **)&base, cap, hipHostMallocDefault) != hipSuccess) {
                base = nullptr;
                cap = 0;
                return nullptr;
            }
            used = bytes;
            return base;
        }
        used = need + bytes;
        return base + need;
    }
};
inline PinStage &pin_stage() {
    static thread_local PinStage p;
    return p;
}
#define TPA_STAGE_CHECK(ptr)                                                                         \
    do {                                                                                             \
        if ((ptr) == nullptr) {                                                                      \
            snprintf(tpa_errbuf, sizeof(tpa_errbuf), "%s:%d: pinned staging allocation failed", __FILE__, __LINE__); \
            return TPA_E_NOMEM;                                                                      \
        }                                                                                            \
    } while (0)

__global__ void post_words_kernel(unsigned int *__restrict__ a, int na, const int *__restrict__ b, unsigned int *__restrict__ host, int zero_after) {
    for (int i = 0; i < na; ++i) {
        host[i] = a[i];
        if (zero_after) a[i] = 0u;      // the counters start the next sweep at zero without a separate memset
    }
    if (b != nullptr) host[na] = (unsigned int)b[0];
    __threadfence_system();
}

template <class T>
inline void stage_put(char *stage, int64_t off0, int64_t off, const std::vector<T> &v) {
    if (!v.empty()) memcpy(stage + (off - off0), v.data(), v.size() * sizeof(T));
}

template <bool CPLX>
int svd_run(const Layout &lay, int n_jobs, const void *a_base, void *u_base, double *s_dev,
            void *vh_base, char *work, int max_sweeps, int *sweeps_done, hipStream_t st, double rho) {
    double *W = (double *)(work + lay.off_w);
    double *G = (double *)(work + lay.off_g);
    double *sig = (double *)(work + lay.off_sig);
    int64_t *perm = (int64_t *)(work + lay.off_perm);
    SvdJob *jobs = (SvdJob *)(work + lay.off_jobs);
    int2 *rows = (int2 *)(work + lay.off_rows);
    int2 *pairs = (int2 *)(work + lay.off_pairs);
    unsigned int *cnt = (unsigned int *)(work + lay.off_cnt);
    BEntry *bent = (BEntry *)(work + lay.off_bent);
    double *gpart = (double *)(work + lay.off_gpart);
    int2 *wpairs = (int2 *)(work + lay.off_wpairs);
    // the activity-driven Gram-only sweeps will run (the same terms as `use_gonly` / `use_dyn` below) -> identity-row skip of the sweep-end product
    const bool dyn_path = !CPLX && !tpa_svd_force_pairwise && tpa_svd_b32 && !lay.b32_pairs.empty() && tpa_svd_gonly && tpa_svd_predict_convergence &&
                          lay.ref.enabled && !lay.b32_gup.empty() && tpa_svd_fused_rounds && tpa_svd_dyn && tpa_svd_lookahead && !lay.b32_act.empty() &&
                          lay.off_sched != 0;
    const bool rot_skip = dyn_path && tpa_svd_apply_skip && !tpa_svd_direct && lay.n_rot > 0;
    int *rot = rot_skip ? (int *)(work + lay.off_rot) : nullptr;
    {   // all host-built tables in one copy out of the pinned arena
        const int64_t t0 = lay.off_jobs, tbytes = lay.off_tab_end - lay.off_jobs;
        char *stg = pin_stage().take((size_t)tbytes, st);
        TPA_STAGE_CHECK(stg);
        stage_put(stg, t0, lay.off_jobs, lay.jobs);
        stage_put(stg, t0, lay.off_rows, lay.rows);
        stage_put(stg, t0, lay.off_pairs, lay.pairs);
        stage_put(stg, t0, lay.off_bent, lay.bentries);
        stage_put(stg, t0, lay.off_wpairs, lay.wpairs);
        stage_put(stg, t0, lay.off_b32e, lay.b32_entries);
        stage_put(stg, t0, lay.off_b32p, lay.b32_pairs);
        if (!lay.b32_act.empty()) stage_put(stg, t0, lay.off_act_ents, lay.b32_act);
        if (lay.ref.enabled) {
            stage_put(stg, t0, lay.off_rtasks, lay.ref.tasks);
            if (rot_skip) {      // the tasks of Qtot [W | G] learn where the "rotated in this sweep" words are (their pad field; the layout is cached, the work area is not)
                int64_t *tk = reinterpret_cast<int64_t *>(stg + (lay.off_rtasks - t0));
                for (int64_t t = lay.ref.apply.task0, te = t + 2 * (int64_t)lay.jobs.size(); t < te; ++t) tk[8 * t + 7] = (int64_t)(intptr_t)(work + lay.off_rot);
            }
            stage_put(stg, t0, lay.off_rlinks, lay.ref.links);
            stage_put(stg, t0, lay.off_rtiles, lay.ref.tiles);
            stage_put(stg, t0, lay.off_rrt, lay.ref.rtiles);
            stage_put(stg, t0, lay.off_gup, lay.b32_gup);
        }
        TPA_HIP_CHECK(hipMemcpyAsync(work + t0, stg, (size_t)tbytes, hipMemcpyHostToDevice, st));
    }
    const int g_rows = (int)(lay.rows.size() / (NT / 64));
    const int g_pairs = (int)(lay.pairs.size() / (NT / 64));
    if (g_rows == 0) {
        if (sweeps_done) *sweeps_done = 0;
        return 0;
    }
    svd_init_kernel<CPLX><<<g_rows, NT, 0, st>>>(jobs, rows, (const double *)a_base, W, G);
    TPA_LAUNCH_CHECK();
    double *fro2 = (double *)(work + lay.off_fro);
    double *fpart = (double *)(work + lay.off_fpart);
    svd_fro_kernel<CPLX><<<dim3(FRO_PARTS, n_jobs), NT, 0, st>>>(jobs, (const double *)a_base, fpart);
    svd_fro_sum_kernel<<<n_jobs, 64, 0, st>>>(jobs, fpart, fro2);
    TPA_LAUNCH_CHECK();
    int sweep = 0;
    bool converged = (lay.rmax_pad < 2);
    unsigned int *posted = (unsigned int *)pin_stage().take(64, st);      // convergence counters (+ error flag), written by post_words_kernel
    TPA_STAGE_CHECK(posted);
    const bool use_block = !tpa_svd_force_pairwise;
    // fused one-launch round: the spin-waits between sibling workgroups need the whole grid resident (4 workgroups of 38 KB LDS
    // per CU) and every part must fit the register-resident chunk budget.  (Oversubscribing the resident set by 4x / 16x was
    // measured in round 2 and is SLOWER, 4.32 / 4.31 against 4.10 s of SVD per sweep: spinning siblings hold the CUs that the
    // late parts need.)
    const bool use_fused = use_block && !CPLX && tpa_svd_fused_round && lay.max_part_chunks <= 4 * FIT &&
                           (int64_t)lay.bentries.size() <= fused_round_capacity();
    const bool use_wide = use_block && !CPLX && tpa_svd_fused_round && tpa_svd_wide_round && lay.wide_ok;
    const bool use_b32 = use_block && !CPLX && tpa_svd_b32 && !lay.b32_pairs.empty();
    B32Entry *b32e = (B32Entry *)(work + lay.off_b32e);
    B32Pair *b32p = (B32Pair *)(work + lay.off_b32p);
    int *b32f = (int *)(work + lay.off_b32f);
    double *b32g = (double *)(work + lay.off_b32g), *b32q = (double *)(work + lay.off_b32q);
    const bool use_fused_c = use_block && CPLX && tpa_svd_fused_round && lay.max_part_chunks <= 4 * FITC &&
                             (int64_t)lay.bentries.size() <= fused_round_capacity_c();
    unsigned int *pcnt = (unsigned int *)(work + lay.off_pcnt);
    int *perr = (int *)(pcnt + lay.bentries.size());
    unsigned int fused_seq = 0;
    if (use_fused || use_fused_c) TPA_HIP_CHECK(hipMemsetAsync(pcnt, 0, lay.bentries.size() * 4 + 4, st));
    const int rounds = use_b32 ? (int)std::max<int64_t>(lay.nb32_max_pad - 1, 1)
                               : use_block ? (int)std::max<int64_t>(lay.nb_max_pad - 1, 1) : (int)std::max<int64_t>(lay.rmax_pad - 1, 1);
    // 32-row-block path: ONE ROUND OF LOOK-AHEAD.  The host has to see the rotation counters of sweep t before it knows whether
    // sweep t + 1 is needed, and the device used to idle for that round trip (~25 us, 6 - 8 times per call: 6 % of a chi = 512
    // call).  Now round 0 of sweep t + 1 is enqueued BEHIND the posting kernel of sweep t and before the host waits (on an event
    // recorded right after the post, not on the stream); if sweep t turns out to have converged, that one round was superfluous
    // but harmless -- it applies the same kind of sub-threshold rotations a further sweep would.
    auto b32_round = [&](int r) {
        const int full_local = (tpa_svd_cross_only && r > 0) ? 0 : 1;
        svd_b32_gram_kernel<<<(int)lay.b32_entries.size(), NTB, 0, st>>>(jobs, b32e, r, W, b32g);
        b32_solve_launch((int)lay.b32_pairs.size(), st, jobs, b32p, b32g, b32q, b32f, cnt, fro2, rho, full_local, nullptr, r);
        svd_b32_apply_kernel<<<(int)lay.b32_entries.size(), NTB, 0, st>>>(jobs, b32e, r, W, G, b32q, b32f);
    };
    // work areas and GEMM tables of the Gram-only sweeps
    constexpr int ES = CPLX ? 2 : 1;
    const RefTables &rt = lay.ref;
    const int64_t *rtasks = (const int64_t *)(work + lay.off_rtasks), *rlinks = (const int64_t *)(work + lay.off_rlinks);
    const int32_t *rtiles = (const int32_t *)(work + lay.off_rtiles);
    const RefTile *rrt = (const RefTile *)(work + lay.off_rrt);
    double *W2 = (double *)(work + lay.off_w2), *G2 = (double *)(work + lay.off_w2 + (lay.off_g - lay.off_w));
    double *P = (double *)(work + lay.off_rp), *Qm = (double *)(work + lay.off_rq), *Mm = (double *)(work + lay.off_rm);
    const int64_t pstride = (lay.g_elems + 1) / 2 * 2 * ES;
    const int n_rt = (int)rt.rtiles.size();
    auto gemm = [&](const RefTables::Span &sp, const void *A, const void *B, void *C) {
        return tpa_gemm_chain(CPLX ? TPA_C128 : TPA_F64, 1, rtasks, rlinks, rtiles + 4 * sp.tile0, sp.n_tiles, A, B, C, st);
    };
    const int jac_limit = max_sweeps;
    double *Wc = W, *Gc = G;            // current [W | G] image (the refinement steps ping-pong between two)
    static thread_local hipEvent_t ev_post = nullptr;
    if (ev_post == nullptr) TPA_HIP_CHECK(hipEventCreateWithFlags(&ev_post, hipEventDisableTiming));
    // ---- Gram-only sweeps (tpa_svd_b32.inc): needs the predicted-convergence rule (the stopping decision must not rest on an
    //      updated Gram matrix alone) and the GEMM tables of the refinement layout
    //      Complex data (tpa_svd_b32c.inc): the Gram-only sweep is the ONLY 32-row-block path (two launches per round).
    const bool use_gonly = (CPLX ? (use_block && tpa_svd_b32 && !lay.b32_pairs.empty()) : use_b32) && tpa_svd_gonly &&
                           tpa_svd_predict_convergence && lay.ref.enabled && !lay.b32_gup.empty();
    const int rounds_g = (int)std::max<int64_t>(lay.nb32_max_pad - 1, 1);
    bool direct = false, direct_started = false;      // tpa_svd_direct: two-sided iteration on the Hermitian input itself
    const double *direct_s = nullptr;                  // where its S ended up
    tpa_svd_direct_used = 0;
    if (use_gonly && !converged && jac_limit > 0) {
        const B32GUp *gup = (const B32GUp *)(work + lay.off_gup);
        double *Wn = W2, *Gn = G2;
        int rc_g = 0;
        auto g_begin = [&]() {        // S = W W^T (both triangles) -> Mm,  Qtot = 1 -> Qm
            if (direct) {             // S = W itself, once; later sweeps go on with the S of the previous one
                if (!direct_started) eigh_direct_init_kernel<CPLX><<<g_rows, NT, 0, st>>>(jobs, rows, Wc, Mm, Qm);
                direct_started = true;
                return;
            }
            if (int rc = gemm(rt.gram, Wc, Wc, P)) rc_g = rc;
            ref_nsm_kernel<CPLX><<<n_rt, NTM, 0, st>>>(jobs, rrt, P, rt.nsplit_g, pstride, Mm, 0.0, -1.0, 1.0, nullptr, Qm);
        };
        const bool fused = !CPLX && tpa_svd_fused_rounds;
        const bool overlap_c = CPLX && tpa_svd_overlap_c;
        {   // the direct iteration exists for the two default round drivers: complex (two launches per round, S in place) and the
            // real activity-driven one; every job must be square (tpa_eigh_batch's are)
            const bool dyn_ok = fused && tpa_svd_dyn && tpa_svd_lookahead && !lay.b32_act.empty() && lay.off_sched != 0;
            bool square = true;
            for (const SvdJob &J : lay.jobs) square = square && (J.R == J.L);
            direct = tpa_svd_direct && square && (CPLX ? !overlap_c : dyn_ok);
        }
        static thread_local hipStream_t st2 = nullptr;
        static thread_local hipEvent_t ev_c[2] = {nullptr, nullptr};
        bool rest_pending = false;
        if (overlap_c && st2 == nullptr) {
            // Measured on the TEBD bonds of config 5 (two 1024 x 1024 blocks), s per step: one stream 4.31, two streams 4.67 (the
            // 68 KB tile workgroups of the second stream fill every CU and the next solve -- 101 KB of LDS, needs an empty CU --
            // queues behind them), second stream from hipExtStreamCreateWithCUMask (3/4 of the CUs, or all of them) 7.5: OFF by default.
            TPA_HIP_CHECK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
            TPA_HIP_CHECK(hipEventCreateWithFlags(&ev_c[0], hipEventDisableTiming));
            TPA_HIP_CHECK(hipEventCreateWithFlags(&ev_c[1], hipEventDisableTiming));
        }
        const int n_pairs = (int)lay.b32_pairs.size(), n_gup = (int)lay.b32_gup.size();
        double *sbuf[2] = {Mm, P};                                       // S_k lives in image k % 2 (P: the split-K partials are dead by then)
        double *qb2[2] = {b32q, b32q + (int64_t)n_pairs * TB * TB};
        int *fb2[2] = {b32f, b32f + n_pairs};
        auto g_round = [&](int r) {
            const int full_local = (tpa_svd_cross_only && r > 0) ? 0 : 1;
            if (CPLX) {
                // complex data: solve(r) -> [tiles the solves of round r + 1 read] on the call's stream, the other tiles on a second
                // stream beside solve(r + 1); the last round of a sweep only has to bring its transforms into Qtot
                const bool last = (r == rounds_g - 1) && !direct;      // (direct: the next sweep continues on this S)
                double2 *qbr = (double2 *)b32q + (int64_t)(r & 1) * n_pairs * TB * TB;
                int *fbr = b32f + (r & 1) * n_pairs;
                svd_b32_solve_c_kernel<<<n_pairs, NTSC, 0, st>>>(b32p, qbr, fbr, cnt, fro2, rho, full_local, (const double2 *)Mm, r);
                const B32GUp *gl = last ? gup + lay.n_gup_s : gup;
                const int nl = last ? n_gup - lay.n_gup_s : n_gup;
                if (!overlap_c) {
                    svd_b32_gupdate_c_kernel<<<nl, NTB, 0, st>>>(gl, r, (double2 *)Mm, (double2 *)Qm, qbr, fbr, 0);
                    return;
                }
                if (hipEventRecord(ev_c[0], st) != hipSuccess || hipStreamWaitEvent(st2, ev_c[0], 0) != hipSuccess) rc_g = 999;
                if (rest_pending && hipStreamWaitEvent(st, ev_c[1], 0) != hipSuccess) rc_g = 999;      // tiles of round r - 1
                if (!last) svd_b32_gupdate_c_kernel<<<lay.n_gup_s, NTB, 0, st>>>(gup, r, (double2 *)Mm, (double2 *)Qm, qbr, fbr, 1);
                svd_b32_gupdate_c_kernel<<<nl, NTB, 0, st2>>>(gl, r, (double2 *)Mm, (double2 *)Qm, qbr, fbr, last ? 0 : 2);
                if (hipEventRecord(ev_c[1], st2) != hipSuccess) rc_g = 999;
                rest_pending = true;
                return;
            }
            if (!fused) {
                b32_solve_launch(n_pairs, st, jobs, b32p, b32g, b32q, b32f, cnt, fro2, rho, full_local, Mm, r);
                svd_b32_gupdate_kernel<<<n_gup, NTB, 0, st>>>(gup, r, Mm, Qm, b32q, b32f);
                return;
            }
            // one launch per round: solve(r) on S_r formed from S_(r-1) and the transforms of round r - 1, + the tiles of update(r - 1)
            if (r == 0)
                svd_b32_solve2_kernel<<<n_pairs, NTS3, 0, st>>>(jobs, b32p, b32g, qb2[0], fb2[0], cnt, fro2, rho, full_local, sbuf[0], 0);
            else
                svd_b32_round_kernel<<<n_pairs + n_gup, NTS3, 0, st>>>(jobs, b32p, n_pairs, gup, r, sbuf[(r - 1) & 1], sbuf[r & 1], Qm,
                                                                       qb2[(r - 1) & 1], qb2[r & 1], fb2[(r - 1) & 1], fb2[r & 1], cnt, fro2,
                                                                       rho, full_local);
            if (r == rounds_g - 1)    // the transforms of the last round still have to reach Qtot (its S tiles are never read)
                svd_b32_gupdate_kernel<<<n_gup - lay.n_gup_s, NTB, 0, st>>>(gup + lay.n_gup_s, r, sbuf[r & 1], Qm, qb2[r & 1], fb2[r & 1]);
        };
        auto g_end = [&]() {          // [W | G] <- Qtot [W | G] into the other image
            if (direct) return;       // (Qtot accumulates over the whole call)
            if (rest_pending) {
                if (hipStreamWaitEvent(st, ev_c[1], 0) != hipSuccess) rc_g = 999;
                rest_pending = false;
            }
            if (int rc = gemm(rt.apply, Qm, Wc, Wn)) rc_g = rc;
            std::swap(Wc, Wn);
            std::swap(Gc, Gn);
        };
        // ---- round 6: DYNAMIC schedule (real data, one launch per round).  The exact Gram matrix of a sweep start says which block
        //      pairs need rotations at all (b32_activity_kernel); only those are scheduled (b32_job_rounds), and a sweep that starts
        //      without "big" pairs is the last one.  With the floor of the stopping rule on the smaller row two thirds of the block
        //      pairs of a chi = 2048 theta are never active (profiles/r06_stopping_rule_emulation.txt).
        const bool use_dyn = fused && tpa_svd_dyn && tpa_svd_lookahead && !lay.b32_act.empty() && lay.off_sched != 0;
        if (rot_skip && !use_dyn) {
            snprintf(tpa_errbuf, sizeof(tpa_errbuf), "tpa_svd_batch: internal error (identity-row skip without the activity-driven rounds)");
            return TPA_E_BADARG;
        }
        if (use_dyn) {
            B32Sched *sched_dev = (B32Sched *)(work + lay.off_sched);
            int *act_dev = (int *)(work + lay.off_act);
            const B32Act *act_ents = (const B32Act *)(work + lay.off_act_ents);
            const int n_act_ents = (int)lay.b32_act.size(), n_jobs_l = (int)lay.jobs.size();
            const int max_rounds = (int)lay.nb32_max_pad + 2;
            int *act_host = (int *)pin_stage().take((size_t)lay.n_act * 4, st);
            TPA_STAGE_CHECK(act_host);
            B32Sched *sched_stage = (B32Sched *)pin_stage().take((size_t)max_rounds * n_pairs * sizeof(B32Sched), st);
            TPA_STAGE_CHECK(sched_stage);
            std::vector<std::vector<B32Matching>> job_rounds(n_jobs_l);      // rounds >= 1 of every job for the sweep at hand
            std::vector<int> pos;                                            // block -> 2 * global pair + half in the previous round
            // entries of round r (r >= 1: from job_rounds, padded by repeating the last matching; r = 0: the static first round)
            auto matching_of = [&](int b, int r, B32Matching &m) {
                const int R = (int)lay.jobs[b].R, NB = (R + BB - 1) / BB, np = ((NB + 1) / 2 * 2) / 2;
                if (r == 0 || job_rounds[b].empty()) {
                    m.resize(np);
                    for (int p = 0; p < np; ++p) b32_host_pair_of(R, p, 0, m[p].first, m[p].second);
                } else
                    m = job_rounds[b][std::min<int>(r - 1, (int)job_rounds[b].size() - 1)];
            };
            auto fill_round = [&](int r) {
                B32Matching m, mp;
                for (int b = 0; b < n_jobs_l; ++b) {
                    const int first = lay.b32_first_pair[b];
                    matching_of(b, r, m);
                    if (r > 0) {
                        matching_of(b, r - 1, mp);
                        pos.assign(2 * mp.size() + 2, -1);
                        for (int p = 0; p < (int)mp.size(); ++p) {
                            pos[mp[p].first] = 2 * (first + p);
                            pos[mp[p].second] = 2 * (first + p) + 1;
                        }
                    }
                    for (int p = 0; p < (int)m.size(); ++p) {
                        B32Sched &E = sched_stage[(size_t)r * n_pairs + first + p];
                        E.bi = m[p].first;
                        E.bj = m[p].second;
                        if (r > 0) {
                            E.srcA = pos[E.bi];
                            E.srcB = pos[E.bj];
                            const std::pair<int, int> &sa = mp[(E.srcA >> 1) - first], &sb = mp[(E.srcB >> 1) - first];
                            E.ax = sa.first;
                            E.ay = sa.second;
                            E.bx = sb.first;
                            E.by = sb.second;
                        } else {
                            E.srcA = E.srcB = E.bx = E.by = -1;
                            E.ax = lay.b32_act_off[b];          // round 0: where the job's activity words are (svd_b32_solve2_kernel)
                            E.ay = 2 * (int)m.size();           // NBp
                        }
                    }
                }
            };
            fill_round(0);
            TPA_HIP_CHECK(hipMemsetAsync(act_dev, 0, (size_t)lay.n_act * 4, st));      // (only the upper triangles of real blocks are ever written)
            TPA_HIP_CHECK(hipMemcpyAsync(sched_dev, sched_stage, (size_t)n_pairs * sizeof(B32Sched), hipMemcpyHostToDevice, st));
            auto sweep_head = [&]() {      // exact Gram matrix, activity of the block pairs, first round: enqueued before the host knows the activity
                if (rot != nullptr && hipMemsetAsync(rot, 0, (size_t)lay.n_rot * 4, st) != hipSuccess) rc_g = 999;
                g_begin();
                b32_activity_kernel<<<n_act_ents, 256, 0, st>>>(jobs, act_ents, sbuf[0], fro2, rho, act_dev);
                if (hipMemcpyAsync(act_host, act_dev, (size_t)lay.n_act * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
                    hipEventRecord(ev_post, st) != hipSuccess)
                    rc_g = 999;
                svd_b32_solve2_kernel<<<n_pairs, NTS3, 0, st>>>(jobs, b32p, b32g, qb2[0], fb2[0], cnt, fro2, rho, 1, sbuf[0], 0, sched_dev,
                                                                tpa_svd_dyn_round0 ? act_dev : nullptr, rot);
            };
            TPA_HIP_CHECK(hipMemsetAsync(cnt, 0, 2 * sizeof(unsigned int), st));
            sweep_head();
            while (sweep < jac_limit) {
                if (rc_g) return rc_g;
                TPA_LAUNCH_CHECK();
                TPA_HIP_CHECK(hipEventSynchronize(ev_post));
                bool any = false, big = false;
                for (int64_t i = 0; i < lay.n_act; ++i) {
                    any |= act_host[i] != 0;
                    big |= (act_host[i] & 2) != 0;
                }
                if (!any) {              // nothing to rotate on exact data (the first round already enqueued finds nothing either)
                    converged = true;
                    break;
                }
                static const int dbg_act = getenv("TPA_SVD_DEBUG_ACT") ? atoi(getenv("TPA_SVD_DEBUG_ACT")) : 0;      // diagnostic: which jobs keep a call's sweeps alive
                if (dbg_act) {
                    std::string line;
                    for (int b = 0; b < n_jobs_l; ++b) {
                        const int R = (int)lay.jobs[b].R, NB = (R + BB - 1) / BB, NBp = (NB + 1) / 2 * 2;
                        int na = 0, nbig = 0;
                        for (int i = 0; i < NBp * NBp; ++i) {
                            na += act_host[lay.b32_act_off[b] + i] != 0;
                            nbig += (act_host[lay.b32_act_off[b] + i] & 2) != 0;
                        }
                        if (na) line += " R" + std::to_string(R) + ":" + std::to_string(na) + (nbig ? "b" : "");
                    }
                    fprintf(stderr, "svd_act sweep %d jobs %d:%s\n", sweep, n_jobs_l, line.c_str());
                }
                int n_r = 1;
                for (int b = 0; b < n_jobs_l; ++b) {
                    b32_job_rounds((int)lay.jobs[b].R, act_host + lay.b32_act_off[b], max_rounds - 1, job_rounds[b]);
                    n_r = std::max(n_r, 1 + (int)job_rounds[b].size());
                }
                for (int r = 1; r < n_r; ++r) fill_round(r);
                if (n_r > 1)
                    TPA_HIP_CHECK(hipMemcpyAsync(sched_dev + n_pairs, sched_stage + n_pairs, (size_t)(n_r - 1) * n_pairs * sizeof(B32Sched),
                                                 hipMemcpyHostToDevice, st));
                for (int r = 1; r < n_r; ++r)
                    svd_b32_round_kernel<<<n_pairs + n_gup, NTS3, 0, st>>>(jobs, b32p, n_pairs, gup, r, sbuf[(r - 1) & 1], sbuf[r & 1], Qm,
                                                                           qb2[(r - 1) & 1], qb2[r & 1], fb2[(r - 1) & 1], fb2[r & 1], cnt, fro2,
                                                                           rho, 0, sched_dev + (size_t)r * n_pairs, sched_dev + (size_t)(r - 1) * n_pairs, rot);
                {   // the transforms of the last round still have to reach Qtot -- and, direct iteration, S (in place: S_(rl+1) in image rl & 1)
                    const int rl = n_r - 1;
                    if (direct) {
                        svd_b32_gupdate_kernel<<<n_gup, NTB, 0, st>>>(gup, rl, sbuf[rl & 1], Qm, qb2[rl & 1], fb2[rl & 1], sched_dev + (size_t)rl * n_pairs);
                        if (rl & 1) std::swap(sbuf[0], sbuf[1]);       // the next sweep starts from sbuf[0]
                    } else
                    svd_b32_gupdate_kernel<<<n_gup - lay.n_gup_s, NTB, 0, st>>>(gup + lay.n_gup_s, rl, sbuf[rl & 1], Qm, qb2[rl & 1], fb2[rl & 1],
                                                                                sched_dev + (size_t)rl * n_pairs);
                }
                g_end();
                ++sweep;
                tpa_svd_dyn_rounds += n_r;
                tpa_svd_dyn_rounds_static += rounds_g;
                ++tpa_svd_dyn_sweeps;
                if (!big && !tpa_svd_strict) {      // the sweep started without big pairs: every rotation it made converges its pair (predicted convergence)
                    converged = true;
                    break;
                }
                if (sweep < jac_limit) sweep_head();
            }
            if (rc_g) return rc_g;
        } else {
        TPA_HIP_CHECK(hipMemsetAsync(cnt, 0, 2 * sizeof(unsigned int), st));
        g_begin();
        g_round(0);
        while (!converged && sweep < jac_limit) {
            for (int r = 1; r < rounds_g; ++r) g_round(r);
            g_end();
            post_words_kernel<<<1, 1, 0, st>>>(cnt, 2, nullptr, posted, 1);
            TPA_HIP_CHECK(hipEventRecord(ev_post, st));
            if (tpa_svd_lookahead && sweep + 1 < jac_limit) {      // look-ahead: Gram matrix and first round of the next sweep (they touch S,
                g_begin();                                         // Qtot and the pair transforms only: harmless if this sweep was the last)
                g_round(0);
            }
            if (rc_g) return rc_g;
            TPA_LAUNCH_CHECK();
            TPA_HIP_CHECK(hipEventSynchronize(ev_post));
            ++sweep;
            converged = (posted[0] == 0) || (posted[1] == 0 && !tpa_svd_strict);
            if (!tpa_svd_lookahead && !converged && sweep < jac_limit) {
                g_begin();
                g_round(0);
            }
        }
        if (rest_pending) TPA_HIP_CHECK(hipStreamWaitEvent(st, ev_c[1], 0));      // (look-ahead round of a sweep that was not needed)
        }
        if (direct) direct_s = CPLX ? Mm : sbuf[0];
    } else
    if (use_b32 && tpa_svd_lookahead && !converged && jac_limit > 0) {
        TPA_HIP_CHECK(hipMemsetAsync(cnt, 0, 2 * sizeof(unsigned int), st));
        b32_round(0);
        while (!converged && sweep < jac_limit) {
            for (int r = 1; r < rounds; ++r) b32_round(r);
            post_words_kernel<<<1, 1, 0, st>>>(cnt, 2, nullptr, posted, 1);
            TPA_HIP_CHECK(hipEventRecord(ev_post, st));
            if (sweep + 1 < jac_limit) b32_round(0);          // look-ahead: first round of the next sweep
            TPA_LAUNCH_CHECK();
            TPA_HIP_CHECK(hipEventSynchronize(ev_post));
            ++sweep;
            converged = (posted[0] == 0) || (tpa_svd_predict_convergence && !tpa_svd_strict && posted[1] == 0);
        }
    }
    while (!converged && sweep < jac_limit) {
        TPA_HIP_CHECK(hipMemsetAsync(cnt, 0, 2 * sizeof(unsigned int), st));
        for (int r = 0; r < rounds; ++r) {
            const int full_local = (tpa_svd_cross_only && r > 0) ? 0 : 1;
            if (use_b32) {
                b32_round(r);
            } else if (use_fused_c) {
                ++fused_seq;
                svd_round_fused_kernel_c<<<(int)lay.bentries.size(), NTG, 0, st>>>(jobs, bent, r, (double2 *)W, (double2 *)G, gpart, pcnt, fused_seq, cnt,
                                                                                  fro2, rho, tpa_svd_local_sweeps, full_local, perr);
            } else if (use_block && CPLX) {
                svd_gram_part_kernel_c<<<(int)lay.bentries.size(), NTG, 0, st>>>(jobs, bent, r, (const double2 *)W, gpart);
                svd_solve_apply_kernel_c<<<(int)lay.bentries.size(), NTG, 0, st>>>(jobs, bent, r, (double2 *)W, (double2 *)G, gpart, cnt, fro2, rho, tpa_svd_local_sweeps, full_local);
            } else if (use_wide) {
                svd_round_wide_kernel<<<(int)lay.wpairs.size(), NTW, 0, st>>>(jobs, wpairs, r, W, G, cnt, fro2, rho, tpa_svd_local_sweeps, full_local);
            } else if (use_fused) {
                ++fused_seq;
                svd_round_fused_kernel<<<(int)lay.bentries.size(), NTG, 0, st>>>(jobs, bent, r, W, G, gpart, pcnt, fused_seq, cnt, fro2, rho,
                                                                                tpa_svd_local_sweeps, full_local, perr);
            } else if (use_block) {
                svd_gram_part_kernel<<<(int)lay.bentries.size(), NTG, 0, st>>>(jobs, bent, r, W, gpart);
                svd_solve_apply_kernel<<<(int)lay.bentries.size(), NTG, 0, st>>>(jobs, bent, r, W, G, gpart, cnt, fro2, rho, tpa_svd_local_sweeps, full_local);
            } else
                svd_round_kernel<CPLX><<<g_pairs, NT, 0, st>>>(jobs, pairs, r, W, G, cnt, fro2, rho);
        }
        TPA_LAUNCH_CHECK();
        unsigned int h2[2] = {0, 0};
        int herr = 0;
        {
            const bool with_err = !use_b32 && ((use_fused && !use_wide) || use_fused_c);
            post_words_kernel<<<1, 1, 0, st>>>(cnt, 2, with_err ? perr : nullptr, posted, 0);
            TPA_HIP_CHECK(hipStreamSynchronize(st));
            h2[0] = posted[0];
            h2[1] = posted[1];
            herr = with_err ? (int)posted[2] : 0;
        }
        if (herr) {
            snprintf(tpa_errbuf, sizeof(tpa_errbuf), "tpa_svd_batch: sibling workgroups of a fused Jacobi round lost each other (spin limit)");
            return TPA_E_NOCONV;
        }
        ++sweep;
        converged = (h2[0] == 0) || (tpa_svd_predict_convergence && !tpa_svd_strict && h2[1] == 0);
    }
    W = Wc;
    G = Gc;
    if (sweeps_done) *sweeps_done = sweep;
    if (direct && direct_s != nullptr) {      // eigenvalues (+ shift) = diag(S); the accumulated transform stands in for G
        eigh_direct_diag_kernel<CPLX><<<g_rows, NT, 0, st>>>(jobs, rows, direct_s, sig);
        // Qtot is a product of (sweeps x rounds) 64 x 64 transforms: orthonormal to ~1e-13 ... 1e-12 only.  One Newton-Schulz step
        // Q <- (3/2 - 1/2 Q Q^H) Q on the GEMM tables of the Gram-only sweeps (square jobs: the W plane and the R x R planes share their
        // offsets) squares that defect -- three n^3 products per CALL instead of three per sweep.
        eigh_direct_init_kernel<CPLX><<<g_rows, NT, 0, st>>>(jobs, rows, Qm, Wc, Gc);      // Wc <- Qtot (Gc <- 1: any finite data)
        if (int rc = gemm(rt.gram, Wc, Wc, P)) return rc;
        ref_nsm_kernel<CPLX><<<n_rt, NTM, 0, st>>>(jobs, rrt, P, rt.nsplit_g, pstride, Mm, 1.5, 0.5, 1.0, nullptr, nullptr);
        double *Wo = (Wc == W) ? W2 : W;        // the other image
        if (int rc = gemm(rt.apply, Mm, Wc, Wo)) return rc;
        G = Wo;
        tpa_svd_direct_used = 1;
    } else
    svd_norms_kernel<CPLX><<<g_rows, NT, 0, st>>>(jobs, rows, W, sig);
    TPA_LAUNCH_CHECK();
    posted[4] = 0u;     // (mapped host word: set by svd_rank_kernel if a singular value is not finite)
    svd_rank_kernel<<<g_rows, NT, 0, st>>>(jobs, rows, sig, perm, posted + 4);      // descending order, stable: no host round trip
    svd_finish_kernel<CPLX><<<g_rows, NT, 0, st>>>(jobs, rows, perm, W, G, sig, (double *)u_base, s_dev, (double *)vh_base);
    TPA_LAUNCH_CHECK();
    TPA_HIP_CHECK(hipStreamSynchronize(st));  // callers read the results (and the staging arena is reused by the next call)
    if (posted[4]) {
        snprintf(tpa_errbuf, sizeof(tpa_errbuf), "tpa_svd_batch: NaN/Inf in singular values");
        return TPA_E_NAN;
    }
    if (!converged) {
        snprintf(tpa_errbuf, sizeof(tpa_errbuf), "tpa_svd_batch: no convergence in %d sweeps", max_sweeps);
        return TPA_E_NOCONV;
    }
    return 0;
}


// ---------------------------------------------------------------------------------------------------
// host driver of the QRP-preconditioned path (real dtype)
struct QrpLayout {
    std::vector<QrpJob> qjobs;
    std::vector<int64_t> nested_max;   // int64[8] jobs of the largest possible nested problem (r = N)
    int64_t m_max = 0;
    int64_t x_elems = 0, r_elems = 0, c_elems = 0, n_max = 0, tf_blocks = 0, off_tfac = 0, off_tpan = 0;
    int64_t off_x = 0, off_vall = 0, off_rtop = 0, off_ur = 0, off_vhr = 0, off_cn = 0, off_tau = 0, off_sr = 0,
            off_cperm = 0, off_qjobs = 0, off_sjobs = 0, off_state = 0, off_fro = 0, off_fpart = 0, off_nested = 0,
            total = 0;
};

QrpLayout build_qrp_layout(int dtype, const int64_t *jobs_host, int n_jobs) {
    QrpLayout q;
    const int64_t esz = (dtype == TPA_C128) ? 16 : 8;
    for (int b = 0; b < n_jobs; ++b) {
        const int64_t m = jobs_host[8 * b + 1], n = jobs_host[8 * b + 2];
        QrpJob J;
        J.tr = (m < n) ? 1 : 0;
        J.M = std::max(m, n);
        J.N = std::min(m, n);
        J.x_off = q.x_elems;
        J.r_off = q.r_elems;
        J.c_off = q.c_elems;
        J.pad0 = q.tf_blocks;   // first Tf block of this job
        J.pad1 = 0;
        q.tf_blocks += (J.N + QNB - 1) / QNB;
        q.x_elems += J.M * J.N;
        q.r_elems += J.N * J.N;
        q.c_elems += J.N;
        q.n_max = std::max(q.n_max, J.N);
        q.m_max = std::max(q.m_max, J.M);
        q.qjobs.push_back(J);
        const int64_t nj[8] = {J.r_off, J.N, J.N, J.r_off, J.c_off, J.r_off, 1, 0};
        q.nested_max.insert(q.nested_max.end(), nj, nj + 8);
    }
    int64_t o = 0;
    auto take = [&o](int64_t bytes) {
        const int64_t at = o;
        o = align_up(o + bytes, 256);
        return at;
    };
    q.off_x = take(q.x_elems * esz);
    q.off_vall = take(q.x_elems * esz);
    q.off_rtop = take(q.r_elems * esz);
    q.off_ur = take(q.r_elems * esz);
    q.off_vhr = take(q.r_elems * esz);
    q.off_cn = take(q.c_elems * 8);
    q.off_tau = take(q.c_elems * esz);
    q.off_sr = take(q.c_elems * 8);
    q.off_cperm = take(q.c_elems * 8);
    q.off_qjobs = take((int64_t)n_jobs * sizeof(QrpJob));
    q.off_sjobs = take((int64_t)n_jobs * sizeof(SvdJob));
    q.off_state = take((int64_t)n_jobs * sizeof(QrpState));
    q.off_fro = take((int64_t)n_jobs * 8);
    q.off_fpart = take((int64_t)n_jobs * 64 * 8);
    q.off_tfac = take(q.tf_blocks * QNB * QNB * esz);
    q.off_tpan = take((int64_t)n_jobs * PNB * PNB * esz);
    q.off_nested = o;
    o += build_layout(dtype, q.nested_max.data(), n_jobs).total;      // (its size only: not worth a slot of the cache)
    q.total = o;
    return q;
}
std::shared_ptr<const QrpLayout> make_qrp_layout(int dtype, const int64_t *jobs_host, int n_jobs) {
    auto build = [&]() { return build_qrp_layout(dtype, jobs_host, n_jobs); };
    if (!tpa_layout_cache_on) return std::make_shared<const QrpLayout>(build());
    static thread_local LayoutCache<QrpLayout> cache;
    return cache.get(dtype, jobs_host, n_jobs, build);
}

constexpr int64_t QRP_MIN_DIM = 32;        // below this the plain Jacobi path is launch-cheaper
constexpr double QRP_RANK_TOL = 1.0e-15;   // residual column norm <= tol * ||A||_F  ->  numerical rank reached
constexpr int QRP_POLL = 8;                // steps between host polls of the "all blocks finished" state

// pinned ring of state snapshots for the non-blocking "all blocks finished?" test of svd_run_qrp (one per host thread)
struct QrpPoll {
    static constexpr int SLOTS = 2;      // the host runs at most SLOTS * QRP_POLL (+ QRP_POLL) steps ahead of the device: superfluous steps stay few
    QrpState *host = nullptr;
    size_t cap = 0;
    hipEvent_t ev[SLOTS] = {};
    bool have_ev = false, busy[SLOTS] = {};
    int at_step[SLOTS] = {}, next = 0;
    int reserve(int n_jobs) {
        if (!have_ev) {
            for (int i = 0; i < SLOTS; ++i) TPA_HIP_CHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
            have_ev = true;
        }
        if (cap < (size_t)n_jobs) {
            drain();
            if (host) TPA_HIP_CHECK(hipHostFree(host));
            cap = (size_t)n_jobs * 2 + 16;
            TPA_HIP_CHECK(hipHostMalloc((void **)&host, sizeof(QrpState) * cap * SLOTS, hipHostMallocDefault));
        }
        return 0;
    }
    bool all_done(int slot, int n_jobs) const {
        for (int b = 0; b < n_jobs; ++b)
            if (!host[(size_t)slot * cap + b].done) return false;
        return true;
    }
    void drain() {      // forget outstanding snapshots (their copies are ordered before anything enqueued later on the stream)
        for (int i = 0; i < SLOTS; ++i)
            if (busy[i]) {
                (void)hipEventSynchronize(ev[i]);
                busy[i] = false;
            }
        next = 0;
    }
};
inline QrpPoll &qrp_poll() {
    static thread_local QrpPoll p;
    return p;
}

template <bool CPLX>
int svd_run_qrp(const Layout &lay, const QrpLayout &q, int n_jobs, const void *a_base, void *u_base, double *s_dev,
                void *vh_base, char *work, int max_sweeps, int *sweeps_done, hipStream_t st, double rho) {
    const int dtype = CPLX ? TPA_C128 : TPA_F64;
    void *X = work + q.off_x, *Vall = work + q.off_vall;
    void *Rtop = work + q.off_rtop, *UR = work + q.off_ur, *VHR = work + q.off_vhr;
    double *cn = (double *)(work + q.off_cn), *SR = (double *)(work + q.off_sr);
    void *tau = work + q.off_tau;
    int64_t *cperm = (int64_t *)(work + q.off_cperm);
    QrpJob *qjobs = (QrpJob *)(work + q.off_qjobs);
    SvdJob *sjobs = (SvdJob *)(work + q.off_sjobs);
    QrpState *state = (QrpState *)(work + q.off_state);
    double *fro2 = (double *)(work + q.off_fro), *fpart = (double *)(work + q.off_fpart);
    {   // qjobs and sjobs are neighbours in the work arena: one staged copy
        const int64_t t0 = q.off_qjobs, tbytes = q.off_sjobs + (int64_t)lay.jobs.size() * sizeof(SvdJob) - q.off_qjobs;
        char *stg = pin_stage().take((size_t)tbytes, st);
        TPA_STAGE_CHECK(stg);
        stage_put(stg, t0, q.off_qjobs, q.qjobs);
        stage_put(stg, t0, q.off_sjobs, lay.jobs);
        TPA_HIP_CHECK(hipMemcpyAsync(work + t0, stg, (size_t)tbytes, hipMemcpyHostToDevice, st));
    }
    svd_fro_kernel<CPLX><<<dim3(FRO_PARTS, n_jobs), NT, 0, st>>>(sjobs, (const double *)a_base, fpart);
    svd_fro_sum_kernel<<<n_jobs, 64, 0, st>>>(sjobs, fpart, fro2);
    { const char *e = getenv("TPA_SVD_SMALL_PANEL"); if (e) tpa_svd_small_panel = atoi(e); }
    const int nmax = (int)q.n_max;
    if (CPLX)
        qrp_init_kernel_c<<<dim3((nmax + 63) / 64, n_jobs), NT, 0, st>>>(qjobs, sjobs, (const cd *)a_base, (cd *)X, cn, cperm, state);
    else
        qrp_init_kernel<<<dim3((nmax + 63) / 64, n_jobs), NT, 0, st>>>(qjobs, sjobs, (const double *)a_base, (double *)X, cn, cperm, state);
    TPA_LAUNCH_CHECK();
    std::vector<QrpState> hstate(n_jobs);
    const double tol2 = QRP_RANK_TOL * QRP_RANK_TOL;
    void *Tpan = work + q.off_tpan;
    bool finished = false;
    for (int k = 0, step = 0;; k += PNB, ++step) {
        if (CPLX) {
            if (q.m_max <= 4 * 256)
                qrp_panel_kernel_c<256, 4><<<n_jobs, 256, 0, st>>>(qjobs, k, (cd *)X, (cd *)Vall, cn, (cd *)tau, cperm, state, fro2, tol2, (cd *)Tpan, 1);
            else
                qrp_panel_kernel_c<256, 8><<<n_jobs, 256, 0, st>>>(qjobs, k, (cd *)X, (cd *)Vall, cn, (cd *)tau, cperm, state, fro2, tol2, (cd *)Tpan, 1);
        } else if (q.m_max <= 8 * 64 && q.n_max <= 8 * 64 && tpa_svd_small_panel)
            // small blocks (chi <= 512, Hubbard ladders): the whole panel in ONE wavefront -- no workgroup barriers, no LDS stage in
            // the reductions (for >= 1000 rows this variant was 1.6x slower, here the barriers are all there is to save)
            qrp_panel_kernel<64, 8><<<n_jobs, 64, 0, st>>>(qjobs, k, (double *)X, (double *)Vall, cn, (double *)tau, cperm, state, fro2, tol2, (double *)Tpan, 1);
        else if (q.m_max <= 8 * 256)
            qrp_panel_kernel<256, 8><<<n_jobs, 256, 0, st>>>(qjobs, k, (double *)X, (double *)Vall, cn, (double *)tau, cperm, state, fro2, tol2, (double *)Tpan, 1);
        else if (q.m_max <= 16 * 256)
            qrp_panel_kernel<256, 16><<<n_jobs, 256, 0, st>>>(qjobs, k, (double *)X, (double *)Vall, cn, (double *)tau, cperm, state, fro2, tol2, (double *)Tpan, 1);
        else
            qrp_panel_kernel<256, 32><<<n_jobs, 256, 0, st>>>(qjobs, k, (double *)X, (double *)Vall, cn, (double *)tau, cperm, state, fro2, tol2, (double *)Tpan, 1);
        if (k >= nmax) break;   // that launch only finalised the states
        const int tiles = (nmax + RCOLS - 1) / RCOLS;   // physical columns; finished ones are skipped inside
        if (CPLX)
            qrp_update_kernel_c<<<dim3(tiles, n_jobs), NTR, 0, st>>>(qjobs, k, (cd *)X, (const cd *)Vall, cn, state, (const cd *)Tpan);
        else
            qrp_update_kernel<<<dim3(tiles, n_jobs), NTR, 0, st>>>(qjobs, k, (double *)X, (const double *)Vall, cn, state, (const double *)Tpan);
        // "all blocks finished?" without draining the queue: every QRP_POLL steps the states are copied to pinned memory behind
        // the launches and the host only LOOKS at a copy once its event has fired, while it keeps enqueueing steps.  Steps that
        // turn out to be superfluous return at their first instruction (state.done).  (Round 3: the blocking poll left the GPU
        // idle for a host round trip ten times per chi = 2048 call.)
        if ((step % QRP_POLL) == QRP_POLL - 1) {
            QrpPoll &P = qrp_poll();
            if (int rc = P.reserve(n_jobs)) return rc;
            const int slot = P.next;
            if (P.busy[slot]) {      // ring full: the oldest copy must have landed before its slot is reused
                TPA_HIP_CHECK(hipEventSynchronize(P.ev[slot]));
                P.busy[slot] = false;
                if (P.all_done(slot, n_jobs)) { finished = true; }
            }
            if (!finished) {
                TPA_HIP_CHECK(hipMemcpyAsync(P.host + (size_t)slot * P.cap, state, n_jobs * sizeof(QrpState), hipMemcpyDeviceToHost, st));
                TPA_HIP_CHECK(hipEventRecord(P.ev[slot], st));
                P.busy[slot] = true;
                P.at_step[slot] = k + PNB;
                P.next = (slot + 1) % QrpPoll::SLOTS;
            }
        }
        {
            QrpPoll &P = qrp_poll();
            for (int sl = 0; sl < QrpPoll::SLOTS && !finished; ++sl)
                if (P.busy[sl] && hipEventQuery(P.ev[sl]) == hipSuccess) {
                    P.busy[sl] = false;
                    if (P.all_done(sl, n_jobs)) finished = true;
                    else if (tpa_svd_rank_cap > 0 && P.at_step[sl] >= tpa_svd_rank_cap) {
                        P.drain();
                        snprintf(tpa_errbuf, sizeof(tpa_errbuf), "tpa_svd_batch: numerical rank above the cap %d", tpa_svd_rank_cap);
                        return TPA_E_RANKCAP;
                    }
                }
        }
        if (finished) break;
    }
    qrp_poll().drain();
    TPA_LAUNCH_CHECK();
    qrp_finish_perm_kernel<<<n_jobs, NT, 0, st>>>(qjobs, state, cn, cperm);
    if (CPLX)
        qrp_extract_kernel_c<<<dim3(64, n_jobs), NT, 0, st>>>(qjobs, state, (const cd *)X, cperm, (cd *)Rtop);
    else
        qrp_extract_kernel<<<dim3(64, n_jobs), NT, 0, st>>>(qjobs, state, (const double *)X, cperm, (double *)Rtop);
    {   // final states and ||A||_F^2 (NaN / Inf in the input: not finite -- the pivot search would silently report rank 0): one wait
        QrpState *pst = (QrpState *)pin_stage().take((size_t)n_jobs * sizeof(QrpState), st);
        double *hfro = (double *)pin_stage().take((size_t)n_jobs * 8, st);
        TPA_STAGE_CHECK(pst);
        TPA_STAGE_CHECK(hfro);
        TPA_HIP_CHECK(hipMemcpyAsync(pst, state, n_jobs * sizeof(QrpState), hipMemcpyDeviceToHost, st));
        TPA_HIP_CHECK(hipMemcpyAsync(hfro, fro2, n_jobs * sizeof(double), hipMemcpyDeviceToHost, st));
        TPA_HIP_CHECK(hipStreamSynchronize(st));
        for (int b = 0; b < n_jobs; ++b) hstate[b] = pst[b];
        for (int b = 0; b < n_jobs; ++b)
            if (!std::isfinite(hfro[b])) {
                snprintf(tpa_errbuf, sizeof(tpa_errbuf), "tpa_svd_batch: NaN/Inf in block %d", b);
                return TPA_E_NAN;
            }
    }
    std::vector<int64_t> nested;
    int rmax = 0, nn = 0;
    for (int b = 0; b < n_jobs; ++b) {
        const QrpJob &J = q.qjobs[b];
        const int r = hstate[b].rank;
        if (!hstate[b].done || r < 0 || r > J.N) {
            snprintf(tpa_errbuf, sizeof(tpa_errbuf), "tpa_svd_batch: pivoted QR did not finish for block %d", b);
            return TPA_E_NOCONV;
        }
        rmax = std::max(rmax, r);
        if (r == 0) continue;
        const int64_t nj[8] = {J.r_off, r, J.N, J.r_off, J.c_off, J.r_off, 1, lay.jobs[b].fro_bits};   // flag 1: orthogonalise the ROWS of R
        nested.insert(nested.end(), nj, nj + 8);
        ++nn;
    }
    int rc = 0;
    if (sweeps_done) *sweeps_done = 0;
    if (nn > 0) {
        // (the nested job table depends on the numerical ranks of THIS call: not memoised)
        const std::shared_ptr<const Layout> nlay_p = std::make_shared<const Layout>(build_layout(dtype, nested.data(), nn));
        const Layout &nlay = *nlay_p;
        if (nlay.total > q.total - q.off_nested) {
            snprintf(tpa_errbuf, sizeof(tpa_errbuf), "tpa_svd_batch: internal work size mismatch");
            return TPA_E_BADARG;
        }
        rc = svd_run<CPLX>(nlay, nn, Rtop, UR, SR, VHR, work + q.off_nested, max_sweeps, sweeps_done, st, rho);
        if (rc != 0 && rc != TPA_E_NOCONV) return rc;
        void *Tfac = work + q.off_tfac;
        const int nblk = (rmax + QNB - 1) / QNB;
        if (CPLX) {
            qrp_form_t_kernel_c<<<dim3(64, n_jobs), NT, 0, st>>>(qjobs, state, (const cd *)UR, (cd *)X);
            qrp_tfactor_kernel_c<<<dim3(nblk, n_jobs), NTR, 0, st>>>(qjobs, state, (const cd *)Vall, (const cd *)tau, (cd *)Tfac);
            for (int blk = nblk - 1; blk >= 0; --blk)
                qrp_apply_q_block_kernel_c<<<dim3((rmax + RCOLS - 1) / RCOLS, n_jobs), NTR, 0, st>>>(qjobs, state, blk, (cd *)X, (const cd *)Vall, (const cd *)Tfac);
        } else {
            qrp_form_t_kernel<<<dim3(64, n_jobs), NT, 0, st>>>(qjobs, state, (const double *)UR, (double *)X);
            qrp_tfactor_kernel<<<dim3(nblk, n_jobs), NTR, 0, st>>>(qjobs, state, (const double *)Vall, (const double *)tau, (double *)Tfac);
            for (int blk = nblk - 1; blk >= 0; --blk)
                qrp_apply_q_block_kernel<<<dim3((rmax + RCOLS - 1) / RCOLS, n_jobs), NTR, 0, st>>>(qjobs, state, blk, (double *)X, (const double *)Vall, (const double *)Tfac);
        }
    }
    if (CPLX)
        qrp_output_kernel_c<<<dim3(128, n_jobs), NT, 0, st>>>(qjobs, sjobs, state, (const cd *)X, SR, (const cd *)VHR, cperm, (cd *)u_base, s_dev, (cd *)vh_base);
    else
        qrp_output_kernel<<<dim3(128, n_jobs), NT, 0, st>>>(qjobs, sjobs, state, (const double *)X, SR, (const double *)VHR, cperm, (double *)u_base, s_dev, (double *)vh_base);
    TPA_LAUNCH_CHECK();
    TPA_HIP_CHECK(hipStreamSynchronize(st));
    return rc;
}


// ---------------------------------------------------------------------------------------------------
// Blocked Householder QR (no pivoting) on the same panel / compact-WY kernels: np.linalg.qr per block for the large
// blocks of the QR-based truncation (np_conserved.py:4190; truncation.py:473-711).  Called from tpa_qr_batch (tpa_qr.hip)
// when min(m, n) >= 32; the one-workgroup kernel there stays for small blocks.
template <bool CPLX>
__global__ __launch_bounds__(NT) void qr_identity_kernel(const QrpJob *__restrict__ jobs, double *__restrict__ T) {
    const QrpJob J = jobs[blockIdx.y];
    const int64_t K = (J.M < J.N) ? J.M : J.N;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < J.M * K; e += (int64_t)gridDim.x * NT) {
        const int64_t i = e / K, j = e - i * K;
        if (CPLX)
            reinterpret_cast<cd *>(T)[J.x_off + e] = cd{(i == j) ? 1.0 : 0.0, 0.0};
        else
            T[J.x_off + e] = (i == j) ? 1.0 : 0.0;
    }
}

template <bool CPLX>
__global__ __launch_bounds__(NT) void qr_copy_q_kernel(const QrpJob *__restrict__ jobs, const int64_t *__restrict__ q_offs,
                                                       const double *__restrict__ T, double *__restrict__ Q) {
    const QrpJob J = jobs[blockIdx.y];
    const int64_t K = (J.M < J.N) ? J.M : J.N, qo = q_offs[blockIdx.y];
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < J.M * K; e += (int64_t)gridDim.x * NT) {
        if (CPLX)
            reinterpret_cast<cd *>(Q)[qo + e] = reinterpret_cast<const cd *>(T)[J.x_off + e];
        else
            Q[qo + e] = T[J.x_off + e];
    }
}

#include "tpa_qr_la.inc"

template <bool CPLX>
int qr_run_wy(const int64_t *jobs_host, int n_jobs, const void *a_base, void *q_base, void *r_base, hipStream_t st) {
    const int64_t esz = CPLX ? 16 : 8;
    std::vector<QrpJob> qj(n_jobs);
    std::vector<SvdJob> sj(n_jobs);
    std::vector<int64_t> qoffs(n_jobs);
    int64_t x_elems = 0, c_elems = 0, tf_blocks = 0, nmax = 0, kmax = 0, mmax = 0;
    for (int b = 0; b < n_jobs; ++b) {
        const int64_t *j = jobs_host + 8 * b;
        QrpJob &J = qj[b];
        J.tr = 0;
        J.M = j[1];
        J.N = j[2];
        J.x_off = x_elems;
        J.r_off = j[4];
        J.c_off = c_elems;
        J.pad0 = tf_blocks;
        J.pad1 = 0;
        const int64_t K = std::min(J.M, J.N);
        tf_blocks += (K + QNB - 1) / QNB;
        x_elems += J.M * J.N;
        c_elems += J.N;
        nmax = std::max(nmax, J.N);
        kmax = std::max(kmax, K);
        mmax = std::max(mmax, J.M);
        qoffs[b] = j[3];
        SvdJob S{};
        S.a_off = j[0];
        S.m = J.M;
        S.n = J.N;
        sj[b] = S;
    }
    int64_t o = 0;
    auto take = [&o](int64_t bytes) {
        const int64_t at = o;
        o = align_up(o + bytes, 256);
        return at;
    };
    const int64_t off_x = take(x_elems * esz), off_v = take(x_elems * esz), off_tau = take(c_elems * esz),
                  off_cn = take(c_elems * 8), off_cperm = take(c_elems * 8), off_qj = take(n_jobs * sizeof(QrpJob)),
                  off_sj = take(n_jobs * sizeof(SvdJob)), off_state = take(n_jobs * sizeof(QrpState)),
                  off_fro = take(n_jobs * 8), off_qo = take(n_jobs * 8), off_tfac = take(tf_blocks * QNB * QNB * esz),
                  off_tpan = take(2 * (int64_t)n_jobs * PNB * PNB * esz);      // (two slots: the look-ahead path ping-pongs)
    char *work = nullptr;
    TPA_HIP_CHECK(hipMallocAsync((void **)&work, (size_t)o, st));
    void *X = work + off_x, *Vall = work + off_v, *tau = work + off_tau, *Tfac = work + off_tfac, *Tpan = work + off_tpan;
    double *cn = (double *)(work + off_cn), *fro2 = (double *)(work + off_fro);
    int64_t *cperm = (int64_t *)(work + off_cperm), *qo_dev = (int64_t *)(work + off_qo);
    QrpJob *qjobs = (QrpJob *)(work + off_qj);
    SvdJob *sjobs = (SvdJob *)(work + off_sj);
    QrpState *state = (QrpState *)(work + off_state);
    int rc = 0;
    auto fail = [&](hipError_t e) {
        snprintf(tpa_errbuf, sizeof(tpa_errbuf), "tpa_qr_batch: %s", hipGetErrorString(e));
        rc = (int)e;
    };
    hipError_t e;
    if ((e = hipMemcpyAsync(qjobs, qj.data(), n_jobs * sizeof(QrpJob), hipMemcpyHostToDevice, st)) != hipSuccess) fail(e);
    if (!rc && (e = hipMemcpyAsync(sjobs, sj.data(), n_jobs * sizeof(SvdJob), hipMemcpyHostToDevice, st)) != hipSuccess) fail(e);
    if (!rc && (e = hipMemcpyAsync(qo_dev, qoffs.data(), n_jobs * 8, hipMemcpyHostToDevice, st)) != hipSuccess) fail(e);
    if (!rc && (e = hipMemsetAsync(fro2, 0, n_jobs * 8, st)) != hipSuccess) fail(e);
    if (!rc) {
        const dim3 gcol((unsigned)((nmax + 63) / 64), n_jobs);
        if (CPLX)
            qrp_init_kernel_c<<<gcol, NT, 0, st>>>(qjobs, sjobs, (const cd *)a_base, (cd *)X, cn, cperm, state);
        else
            qrp_init_kernel<<<gcol, NT, 0, st>>>(qjobs, sjobs, (const double *)a_base, (double *)X, cn, cperm, state);
        const int tiles = (int)((nmax + RCOLS - 1) / RCOLS);
        bool tall = true;      // (wide blocks have columns beyond the last panel that belong to neither role of the fused step)
        for (int b = 0; b < n_jobs; ++b) tall = tall && (qj[b].M >= qj[b].N);
        const bool lookahead = !CPLX && tpa_qr_lookahead && tall && mmax <= (int64_t)QLA_RPT * NTQ;
        if (lookahead) {
            // one launch per panel: workgroup 0 of every block updates + factorises panel k + 1, the others apply panel k behind it
            for (int k = -PNB; k < kmax; k += PNB) {
                const int64_t j_first = (k < 0) ? 0 : ((int64_t)k + 2 * PNB) / RCOLS * RCOLS;
                const int upd = (k < 0 || j_first >= nmax) ? 0 : (int)((nmax - j_first + RCOLS - 1) / RCOLS);
                qr_la_step_kernel<<<dim3(1 + upd, n_jobs), NTQ, 0, st>>>(qjobs, k, (double *)X, (double *)Vall, (double *)tau, (double *)Tpan,
                                                                        n_jobs, j_first);
            }
            // (the states: rank = min(m, n) for every block -- the two-kernel path's final panel launch does exactly that)
            qrp_panel_kernel<256, 8><<<n_jobs, 256, 0, st>>>(qjobs, (int)((kmax + PNB - 1) / PNB * PNB), (double *)X, (double *)Vall, cn, (double *)tau,
                                                            cperm, state, fro2, 0.0, (double *)Tpan, 0);
        }
        for (int k = 0; !lookahead; k += PNB) {
            if (CPLX) {
                if (mmax <= 4 * 256)
                    qrp_panel_kernel_c<256, 4><<<n_jobs, 256, 0, st>>>(qjobs, k, (cd *)X, (cd *)Vall, cn, (cd *)tau, cperm, state, fro2, 0.0, (cd *)Tpan, 0);
                else
                    qrp_panel_kernel_c<256, 8><<<n_jobs, 256, 0, st>>>(qjobs, k, (cd *)X, (cd *)Vall, cn, (cd *)tau, cperm, state, fro2, 0.0, (cd *)Tpan, 0);
            } else if (mmax <= 8 * 256)
                qrp_panel_kernel<256, 8><<<n_jobs, 256, 0, st>>>(qjobs, k, (double *)X, (double *)Vall, cn, (double *)tau, cperm, state, fro2, 0.0, (double *)Tpan, 0);
            else if (mmax <= 16 * 256)
                qrp_panel_kernel<256, 16><<<n_jobs, 256, 0, st>>>(qjobs, k, (double *)X, (double *)Vall, cn, (double *)tau, cperm, state, fro2, 0.0, (double *)Tpan, 0);
            else
                qrp_panel_kernel<256, 32><<<n_jobs, 256, 0, st>>>(qjobs, k, (double *)X, (double *)Vall, cn, (double *)tau, cperm, state, fro2, 0.0, (double *)Tpan, 0);
            if (k >= kmax) break;   // that launch only finalised the states (rank = min(m, n))
            if (CPLX)
                qrp_update_kernel_c<<<dim3(tiles, n_jobs), NTR, 0, st>>>(qjobs, k, (cd *)X, (const cd *)Vall, cn, state, (const cd *)Tpan);
            else
                qrp_update_kernel<<<dim3(tiles, n_jobs), NTR, 0, st>>>(qjobs, k, (double *)X, (const double *)Vall, cn, state, (const double *)Tpan);
        }
        const int nblk = (int)((kmax + QNB - 1) / QNB);
        const dim3 gq((unsigned)((kmax + RCOLS - 1) / RCOLS), n_jobs);
        if (CPLX) {
            qrp_extract_kernel_c<<<dim3(64, n_jobs), NT, 0, st>>>(qjobs, state, (const cd *)X, cperm, (cd *)r_base);
            qr_identity_kernel<true><<<dim3(64, n_jobs), NT, 0, st>>>(qjobs, (double *)X);
            qrp_tfactor_kernel_c<<<dim3(nblk, n_jobs), NTR, 0, st>>>(qjobs, state, (const cd *)Vall, (const cd *)tau, (cd *)Tfac);
            for (int blk = nblk - 1; blk >= 0; --blk)
                qrp_apply_q_block_kernel_c<<<gq, NTR, 0, st>>>(qjobs, state, blk, (cd *)X, (const cd *)Vall, (const cd *)Tfac);
            qr_copy_q_kernel<true><<<dim3(64, n_jobs), NT, 0, st>>>(qjobs, qo_dev, (const double *)X, (double *)q_base);
        } else {
            qrp_extract_kernel<<<dim3(64, n_jobs), NT, 0, st>>>(qjobs, state, (const double *)X, cperm, (double *)r_base);
            qr_identity_kernel<false><<<dim3(64, n_jobs), NT, 0, st>>>(qjobs, (double *)X);
            qrp_tfactor_kernel<<<dim3(nblk, n_jobs), NTR, 0, st>>>(qjobs, state, (const double *)Vall, (const double *)tau, (double *)Tfac);
            for (int blk = nblk - 1; blk >= 0; --blk)
                qrp_apply_q_block_kernel<<<gq, NTR, 0, st>>>(qjobs, state, blk, (double *)X, (const double *)Vall, (const double *)Tfac);
            qr_copy_q_kernel<false><<<dim3(64, n_jobs), NT, 0, st>>>(qjobs, qo_dev, (const double *)X, (double *)q_base);
        }
        if ((e = hipGetLastError()) != hipSuccess) fail(e);
    }
    (void)hipFreeAsync(work, st);
    return rc;
}

}  // namespace

// internal entry used by tpa_qr_batch (tpa_qr.hip)
int tpa_qr_wy_internal(int dtype, const int64_t *jobs_host, int n_jobs, const void *a_base, void *q_base, void *r_base,
                       void *stream) {
    if (dtype == TPA_F64) return qr_run_wy<false>(jobs_host, n_jobs, a_base, q_base, r_base, (hipStream_t)stream);
    return qr_run_wy<true>(jobs_host, n_jobs, a_base, q_base, r_base, (hipStream_t)stream);
}

extern "C" int64_t tpa_svd_worksize(int dtype, const int64_t *jobs_host, int n_jobs) {
    if (n_jobs <= 0) return 256;
    int64_t total = make_layout(dtype, jobs_host, n_jobs)->total;
    total = std::max(total, make_qrp_layout(dtype, jobs_host, n_jobs)->total);
    return total;
}

// diagnostic ring of the last calls of tpa_svd_batch: {rows of the largest block (min(m, n)), its columns, blocks, sweeps,
// pivoted QR used, algorithm switches (b32 | gonly << 1 | refine << 2), wall microseconds of the call, return code}
constexpr int CALL_LOG_N = 8192;
static int64_t tpa_svd_call_log_buf[CALL_LOG_N][8];
static int64_t tpa_svd_call_log_count = 0;
static int tpa_svd_batch_impl(int dtype, const int64_t *jobs_host, int n_jobs, const void *a_base,
                              void *u_base, double *s_dev, void *vh_base, void *work_dev,
                              int64_t work_bytes, int max_sweeps, double tol, int *sweeps_done,
                              void *stream, int *used_qrp);

extern "C" int64_t tpa_svd_call_log(int64_t *out, int64_t max_rows, int reset) {
    const int64_t n = std::min<int64_t>(std::min<int64_t>(tpa_svd_call_log_count, CALL_LOG_N), max_rows);
    const int64_t first = tpa_svd_call_log_count - n;
    for (int64_t i = 0; i < n; ++i) memcpy(out + 8 * i, tpa_svd_call_log_buf[(first + i) % CALL_LOG_N], 64);
    if (reset) tpa_svd_call_log_count = 0;
    return n;
}

extern "C" int tpa_svd_batch(int dtype, const int64_t *jobs_host, int n_jobs, const void *a_base,
                             void *u_base, double *s_dev, void *vh_base, void *work_dev,
                             int64_t work_bytes, int max_sweeps, double tol, int *sweeps_done,
                             void *stream) {
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int used_qrp = 0, sw = 0;
    const int rc = tpa_svd_batch_impl(dtype, jobs_host, n_jobs, a_base, u_base, s_dev, vh_base, work_dev, work_bytes, max_sweeps, tol,
                                      &sw, stream, &used_qrp);
    if (sweeps_done) *sweeps_done = sw;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    int64_t rmax = 0, lmax = 0;
    for (int b = 0; b < n_jobs; ++b) {
        const int64_t r = std::min(jobs_host[8 * b + 1], jobs_host[8 * b + 2]);
        if (r > rmax) {
            rmax = r;
            lmax = std::max(jobs_host[8 * b + 1], jobs_host[8 * b + 2]);
        }
    }
    int64_t *e = tpa_svd_call_log_buf[tpa_svd_call_log_count % CALL_LOG_N];
    e[0] = rmax;
    e[1] = lmax;
    e[2] = n_jobs;
    e[3] = sw;
    e[4] = used_qrp;
    e[5] = tpa_svd_b32 | (tpa_svd_gonly << 1);
    e[6] = (int64_t)(t1.tv_sec - t0.tv_sec) * 1000000 + (t1.tv_nsec - t0.tv_nsec) / 1000;
    e[7] = rc;
    ++tpa_svd_call_log_count;
    return rc;
}

static int tpa_svd_batch_impl(int dtype, const int64_t *jobs_host, int n_jobs, const void *a_base,
                              void *u_base, double *s_dev, void *vh_base, void *work_dev,
                              int64_t work_bytes, int max_sweeps, double tol, int *sweeps_done,
                              void *stream, int *used_qrp) {
    TPA_ARG_CHECK(tol >= -1.0 && tol <= 1.0);
    TPA_ARG_CHECK(dtype == TPA_F64 || dtype == TPA_C128);
    if (n_jobs <= 0) return 0;
    for (int b = 0; b < n_jobs; ++b) TPA_ARG_CHECK(jobs_host[8 * b + 1] > 0 && jobs_host[8 * b + 2] > 0);
    // tol < 0: floor |tol| on the SMALLER row of a pair (svd_needs_rotation; the caller post-processes with the ordered clean-up): the
    // sign goes down to the kernels with the value (svd_floor2)
    pin_stage().reset();      // the previous call on this thread ended with a stream synchronisation
    const std::shared_ptr<const Layout> lay_p = make_layout(dtype, jobs_host, n_jobs);
    const Layout &lay = *lay_p;
    TPA_ARG_CHECK(work_bytes >= lay.total);
    hipStream_t st = (hipStream_t)stream;
    int64_t dim_max = 0;
    for (int b = 0; b < n_jobs; ++b) dim_max = std::max(dim_max, std::max(jobs_host[8 * b + 1], jobs_host[8 * b + 2]));
    // complex: the panel lives in registers as (re, im) pairs -> max(m, n) <= 2048
    if (tpa_svd_use_qrp && !tpa_svd_force_pairwise && lay.rmax_pad >= QRP_MIN_DIM &&
        dim_max <= ((dtype == TPA_F64) ? (int64_t)NTP_MAX * RPT_MAX : (int64_t)2048)) {
        const std::shared_ptr<const QrpLayout> q_p = make_qrp_layout(dtype, jobs_host, n_jobs);
        const QrpLayout &q = *q_p;
        TPA_ARG_CHECK(work_bytes >= q.total);
        *used_qrp = 1;
        if (dtype == TPA_F64)
            return svd_run_qrp<false>(lay, q, n_jobs, a_base, u_base, s_dev, vh_base, (char *)work_dev, max_sweeps, sweeps_done, st, tol);
        return svd_run_qrp<true>(lay, q, n_jobs, a_base, u_base, s_dev, vh_base, (char *)work_dev, max_sweeps, sweeps_done, st, tol);
    }
    if (dtype == TPA_F64)
        return svd_run<false>(lay, n_jobs, a_base, u_base, s_dev, vh_base, (char *)work_dev, max_sweeps, sweeps_done, st, tol);
    return svd_run<true>(lay, n_jobs, a_base, u_base, s_dev, vh_base, (char *)work_dev, max_sweeps, sweeps_done, st, tol);
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

constexpr int EIGH_PARTS = 32;      // workgroups per job of the elementwise passes (a batched TEBD call holds ~100 blocks of 1024 x 1024)
template <bool CPLX>
__global__ __launch_bounds__(NT) void eigh_fro_kernel(const EighJob *__restrict__ jobs, const double *__restrict__ A, double *__restrict__ fpart) {
    __shared__ double red[NT / 64];
    const EighJob J = jobs[blockIdx.y];
    const int64_t tot = J.n * J.n * (CPLX ? 2 : 1);
    const double *a = A + (CPLX ? 2 : 1) * J.a_off;
    double s = 0;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < tot; e += (int64_t)EIGH_PARTS * NT) s = fma(a[e], a[e], s);
    s = block_sum<NT>(s, red);
    if (threadIdx.x == 0) fpart[blockIdx.y * EIGH_PARTS + blockIdx.x] = s;
}

template <bool CPLX>
__global__ __launch_bounds__(NT) void eigh_shift_kernel(const EighJob *__restrict__ jobs,
                                                        const double *__restrict__ A, const double *__restrict__ fpart,
                                                        double *__restrict__ Ap, double *__restrict__ mu) {
    const EighJob J = jobs[blockIdx.y];
    const int64_t n = J.n;
    const double *a = A + (CPLX ? 2 : 1) * J.a_off;
    double *ap = Ap + (CPLX ? 2 : 1) * J.ap_off;
    double s = 0;
    for (int k = 0; k < EIGH_PARTS; ++k) s += fpart[blockIdx.y * EIGH_PARTS + k];      // fixed order: every workgroup gets the same bits
    // mu = 2 ||A||_F (1 if A == 0): spectrum of A' lies in [||A||_F, 3 ||A||_F] > 0, so every singular
    // vector is well defined and the Jacobi iteration sees a condition number <= 3.
    const double m = (s > 0.0) ? 2.0 * sqrt(s) : 1.0;
    if (threadIdx.x == 0 && blockIdx.x == 0) mu[blockIdx.y] = m;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < n * n; e += (int64_t)EIGH_PARTS * NT) {
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
                                                         double *__restrict__ Wout, double *__restrict__ V, int from_vh) {
    // from_vh (the direct two-sided iteration): `U` is the VH plane of the SVD layout, whose row k is the conjugate of row k of the
    // accumulated transform; the job ran on W = A'^T, so eigenvector k of A' is conj(VH[k, :]).  (The U plane is W / sigma there: unused.)
    const EighJob J = jobs[blockIdx.y];
    const int64_t n = J.n;
    const double m = mu[blockIdx.y];
    if (blockIdx.x == 0)
        for (int64_t j = threadIdx.x; j < n; j += NT) Wout[J.w_off + j] = S[J.s_off + (n - 1 - j)] - m;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < n * n; e += (int64_t)gridDim.x * NT) {
        const int64_t i = e / n, j = e % n;
        const int64_t src = from_vh ? (J.ap_off + (n - 1 - j) * n + i) : (J.ap_off + i * n + (n - 1 - j));
        if (CPLX) {
            double2 v = reinterpret_cast<const double2 *>(U)[src];
            if (from_vh) v.y = -v.y;
            reinterpret_cast<double2 *>(V)[J.v_off + e] = v;
        } else
            V[J.v_off + e] = U[src];
    }
}

struct EighLayout {
    std::vector<EighJob> jobs;
    std::vector<int64_t> svd_jobs;  // int64[8] per job, offsets into the workspace planes
    int64_t mat_elems = 0, s_elems = 0;
    int64_t off_ap = 0, off_u = 0, off_vh = 0, off_s = 0, off_mu = 0, off_fpart = 0, off_jobs = 0, off_svd = 0, total = 0;
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
    lay.off_fpart = o;
    o = align_up(o + (int64_t)n_jobs * EIGH_PARTS * 8, 256);
    lay.off_jobs = o;
    o = align_up(o + (int64_t)n_jobs * sizeof(EighJob), 256);
    lay.off_svd = o;
    o += make_layout(dtype, lay.svd_jobs.data(), n_jobs)->total;
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
    pin_stage().reset();
    EighLayout lay = make_eigh_layout(dtype, jobs_host, n_jobs);
    TPA_ARG_CHECK(work_bytes >= lay.total);
    hipStream_t st = (hipStream_t)stream;
    char *work = (char *)work_dev;
    EighJob *jobs = (EighJob *)(work + lay.off_jobs);
    double *mu = (double *)(work + lay.off_mu);
    double *fpart = (double *)(work + lay.off_fpart);
    TPA_HIP_CHECK(hipMemcpyAsync(jobs, lay.jobs.data(), lay.jobs.size() * sizeof(EighJob), hipMemcpyHostToDevice, st));
    const std::shared_ptr<const Layout> slay_p = make_layout(dtype, lay.svd_jobs.data(), n_jobs);
    const Layout &slay = *slay_p;
    int rc;
    int direct_req = tpa_eigh_direct;
    if (const char *e = getenv("TPA_EIGH_DIRECT")) direct_req = atoi(e) != 0;
    constexpr int EIGH_FIN_PARTS = EIGH_PARTS;
    if (dtype == TPA_F64) {
        eigh_fro_kernel<false><<<dim3(EIGH_PARTS, n_jobs), NT, 0, st>>>(jobs, (const double *)a_base, fpart);
        eigh_shift_kernel<false><<<dim3(EIGH_PARTS, n_jobs), NT, 0, st>>>(jobs, (const double *)a_base, fpart, (double *)(work + lay.off_ap), mu);
        TPA_LAUNCH_CHECK();
        tpa_svd_strict = 1;
        tpa_svd_direct = direct_req;
        rc = svd_run<false>(slay, n_jobs, work + lay.off_ap, work + lay.off_u, (double *)(work + lay.off_s),
                            work + lay.off_vh, work + lay.off_svd, max_sweeps, sweeps_done, st, 0.0);
        tpa_svd_strict = tpa_svd_direct = 0;
        if (rc != 0) return rc;
        const int fv = tpa_svd_direct_used;
        eigh_finish_kernel<false><<<dim3(EIGH_FIN_PARTS, n_jobs), NT, 0, st>>>(jobs, (const double *)(work + (fv ? lay.off_vh : lay.off_u)), (const double *)(work + lay.off_s), mu, w_dev, (double *)v_base, fv);
    } else {
        eigh_fro_kernel<true><<<dim3(EIGH_PARTS, n_jobs), NT, 0, st>>>(jobs, (const double *)a_base, fpart);
        eigh_shift_kernel<true><<<dim3(EIGH_PARTS, n_jobs), NT, 0, st>>>(jobs, (const double *)a_base, fpart, (double *)(work + lay.off_ap), mu);
        TPA_LAUNCH_CHECK();
        tpa_svd_strict = 1;
        tpa_svd_direct = direct_req;
        rc = svd_run<true>(slay, n_jobs, work + lay.off_ap, work + lay.off_u, (double *)(work + lay.off_s),
                           work + lay.off_vh, work + lay.off_svd, max_sweeps, sweeps_done, st, 0.0);
        tpa_svd_strict = tpa_svd_direct = 0;
        if (rc != 0) return rc;
        const int fv = tpa_svd_direct_used;
        eigh_finish_kernel<true><<<dim3(EIGH_FIN_PARTS, n_jobs), NT, 0, st>>>(jobs, (const double *)(work + (fv ? lay.off_vh : lay.off_u)), (const double *)(work + lay.off_s), mu, w_dev, (double *)v_base, fv);
    }
    TPA_LAUNCH_CHECK();
    TPA_HIP_CHECK(hipStreamSynchronize(st));
    return 0;
}

// ---- eigenpairs of a Hermitian block out of its SVD (np_conserved.eigh_batched, real data) ---------------------------------------
// A = U S V^H with A Hermitian: wherever |lambda| is not degenerate between a positive and a negative eigenvalue, v_i = d_i u_i with
// d_i = +/-1, lambda_i = d_i sigma_i and u_i is the eigenvector.  The SVD path (pivoted QR + one-sided Jacobi on the rank-r factor) needs
// ~7 sweeps on the graded, rank-deficient density matrices of the mixer where the Jacobi iteration on the matrix itself needs 35 - 40
// (clusters below the off-diagonal norm converge linearly).  This kernel CHECKS the premise instead of assuming it: per vector
// d_i = sign(Re u_i . v_i) and err_i = sigma_i |v_i - d_i u_i| (the residual of u_i as an eigenvector, |A u_i - lambda_i u_i|);
// lam <- d_i sigma_i, errmax[job] <- max_i err_i.  The caller accepts the result only if errmax is at rounding level and takes the
// two-sided iteration otherwise (+/- pairs, a matrix that is not Hermitian in its upper triangle).
namespace {
struct EigFromSvdJob {  // int64[8]
    int64_t u_off, n, s_off, vh_off, lam_off, pad0, pad1, pad2;
};
template <bool CPLX>
__global__ __launch_bounds__(256) void eigh_from_svd_kernel(const EigFromSvdJob *__restrict__ jobs, const double *__restrict__ U,
                                                            const double *__restrict__ S, const double *__restrict__ VH,
                                                            double *__restrict__ lam, unsigned long long *__restrict__ errmax) {
    constexpr int ES = CPLX ? 2 : 1;
    __shared__ double tr[64][65], ti[CPLX ? 64 : 1][65];
    __shared__ double red[4][64];
    const EigFromSvdJob J = jobs[blockIdx.y];
    const int64_t n = J.n;
    const int i0 = blockIdx.x * 64;
    if (i0 >= n) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = i0 + lane;               // this thread's vector (column of U, row of VH)
    const double *Ub = U + ES * J.u_off, *Vb = VH + ES * J.vh_off;
    double sgn = 1.0;
    double out = 0.0;
    for (int pass = 0; pass < 2; ++pass) {
        double acc = 0.0;
        for (int64_t k0 = 0; k0 < n; k0 += 64) {
            __syncthreads();
            // VH tile: rows i0 .. i0 + 63, columns k0 .. k0 + 63, read along k (coalesced); v_i[k] = conj(VH[i, k])
            for (int e = threadIdx.x; e < 64 * 64; e += 256) {
                const int r = e >> 6, c = e & 63;
                const bool ok = (i0 + r < n) && (k0 + c < n);
                const int64_t g = (int64_t)(i0 + r) * n + k0 + c;
                tr[r][c] = ok ? Vb[ES * g] : 0.0;
                if (CPLX) ti[r][c] = ok ? -Vb[ES * g + 1] : 0.0;
            }
            __syncthreads();
            if (i < n) {
                for (int kk = wave * 16; kk < wave * 16 + 16; ++kk) {
                    const int64_t k = k0 + kk;
                    if (k >= n) break;
                    const double ur = Ub[ES * (k * n + i)], ui = CPLX ? Ub[ES * (k * n + i) + 1] : 0.0;
                    const double vr = tr[lane][kk], vi = CPLX ? ti[lane][kk] : 0.0;
                    if (pass == 0)
                        acc += ur * vr + ui * vi;                     // Re conj(u) . v
                    else {
                        const double dr = vr - sgn * ur, di = vi - sgn * ui;
                        acc += dr * dr + di * di;
                    }
                }
            }
        }
        __syncthreads();
        red[wave][lane] = acc;
        __syncthreads();
        const double tot = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
        if (pass == 0)
            sgn = (tot < 0.0) ? -1.0 : 1.0;
        else
            out = tot;
    }
    double e = 0.0;
    if (i < n && wave == 0) {
        const double sg = S[J.s_off + i];
        lam[J.lam_off + i] = sgn * sg;
        e = sg * sqrt(out);
        if (!(e == e)) e = 1.0e300;
    }
    if (wave == 0) {
        e = wave_max(e);
        if (lane == 0 && e > 0.0) atomicMax(errmax + blockIdx.y, (unsigned long long)__double_as_longlong(e));
    }
}
}  // namespace

/* Eigenvalues (with their signs) of Hermitian blocks from their SVDs + the check that the left singular vectors ARE the eigenvectors.
 * jobs : int64[n_jobs][8] = {u_off, n, s_off, vh_off, lam_off, 0, 0, 0} (HOST); U_b n x n (vectors = columns), VH_b n x n (rows).
 * lam_dev[lam_off + i] = d_i S_i, err_dev[job] = max_i S_i |v_i - d_i u_i| (double; zeroed here).  Asynchronous on `stream`. */
extern "C" int tpa_eigh_from_svd(int dtype, const int64_t *jobs_host, int n_jobs, const void *u_base, const double *s_dev,
                                 const void *vh_base, double *lam_dev, double *err_dev, void *stream) {
    TPA_ARG_CHECK(dtype == TPA_F64 || dtype == TPA_C128);
    if (n_jobs <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    pin_stage().reset();
    EigFromSvdJob *stg = (EigFromSvdJob *)pin_stage().take((size_t)n_jobs * sizeof(EigFromSvdJob), st);
    TPA_STAGE_CHECK(stg);
    int64_t nmax = 0;
    for (int b = 0; b < n_jobs; ++b) {
        const int64_t *j = jobs_host + 8 * b;
        TPA_ARG_CHECK(j[1] > 0);
        stg[b] = EigFromSvdJob{j[0], j[1], j[2], j[3], j[4], 0, 0, 0};
        nmax = std::max(nmax, j[1]);
    }
    EigFromSvdJob *jd = nullptr;
    TPA_HIP_CHECK(hipMallocAsync((void **)&jd, (size_t)n_jobs * sizeof(EigFromSvdJob), st));
    TPA_HIP_CHECK(hipMemcpyAsync(jd, stg, (size_t)n_jobs * sizeof(EigFromSvdJob), hipMemcpyHostToDevice, st));
    TPA_HIP_CHECK(hipMemsetAsync(err_dev, 0, (size_t)n_jobs * 8, st));
    const dim3 grid((unsigned)((nmax + 63) / 64), (unsigned)n_jobs);
    if (dtype == TPA_F64)
        eigh_from_svd_kernel<false><<<grid, 256, 0, st>>>(jd, (const double *)u_base, s_dev, (const double *)vh_base, lam_dev, (unsigned long long *)err_dev);
    else
        eigh_from_svd_kernel<true><<<grid, 256, 0, st>>>(jd, (const double *)u_base, s_dev, (const double *)vh_base, lam_dev, (unsigned long long *)err_dev);
    TPA_LAUNCH_CHECK();
    TPA_HIP_CHECK(hipFreeAsync(jd, st));
    return 0;
}

/* Test hook: 0 = tpa_eigh_batch takes the shift + one-sided SVD route of rounds 1 - 5, 1 (default) = the direct two-sided iteration. */
extern "C" int tpa_eigh_set_direct(int on) {
    tpa_eigh_direct = on ? 1 : 0;
    return 0;
}

extern "C" int tpa_svd_set_rank_cap(int cap) {
    tpa_svd_rank_cap = cap > 0 ? cap : 0;
    return 0;
}

extern "C" int tpa_svd_set_algorithm(int pairwise) {
    tpa_svd_force_pairwise = (pairwise & 1) ? 1 : 0;
    tpa_svd_cross_only = (pairwise & 4) ? 0 : 1;
    tpa_svd_use_qrp = (pairwise & 512) ? 0 : 1;    // bit 9: no pivoted-QR preconditioner
    tpa_svd_fused_round = (pairwise & 2) ? 0 : 1;  // bit 1: two-kernel rounds (gram, then solve + apply)
    tpa_svd_wide_round = (pairwise & 2048) ? 0 : 1;   // bit 11: no one-workgroup-per-pair round (-> fused round with column parts)
    tpa_svd_b32 = ((pairwise & 4096) || (pairwise & 2)) ? 0 : 1;   // bit 12 (or the two-kernel 8-row rounds of bit 1): no 32-row-block rounds
    tpa_svd_lookahead = (pairwise & 8192) ? 0 : 1;     // bit 13: no look-ahead round (the host drains the stream after every sweep)
    tpa_svd_predict_convergence = (pairwise & 1024) ? 0 : 1;   // bit 10: always run the verification sweep (see svd_big_rotation)
    if ((pairwise & 0xf0) || (pairwise & 256)) tpa_svd_local_sweeps = (pairwise >> 4) & 15;   // test hook: local sweeps in bits 4..7 (256 -> 0)
    tpa_svd_fused_rounds = (pairwise & 8388608) ? 0 : 1;    // bit 23: two launches per Gram-only round
    tpa_svd_overlap_c = (pairwise & 16384) ? 1 : 0;   // bit 14: complex Gram-only rounds with the non-urgent tiles on a second stream (off by default)
    tpa_svd_gonly = (pairwise & 1048576) ? 0 : 1;     // bit 20: no Gram-only sweeps (the round-3 rounds: gram, solve, apply on the data)
    tpa_svd_dyn_round0 = (pairwise & 33554432) ? 0 : 1;   // bit 25: the first round of every sweep rotates all 2016 local pairs of every block pair
    tpa_svd_dyn = (pairwise & 16777216) ? 0 : 1;      // bit 24: the full round-robin schedule in every sweep instead of the activity-driven one (round 6)
    return 0;
}

/* Rounds launched by the activity-driven schedule of the Gram-only sweeps against the round-robin count: out = {rounds launched,
 * rounds of the static schedule for the same sweeps, sweeps}; reset != 0 clears the counters. */
extern "C" int tpa_svd_dyn_stats(int64_t *out, int reset) {
    if (out) {
        out[0] = tpa_svd_dyn_rounds;
        out[1] = tpa_svd_dyn_rounds_static;
        out[2] = tpa_svd_dyn_sweeps;
    }
    if (reset) tpa_svd_dyn_rounds = tpa_svd_dyn_rounds_static = tpa_svd_dyn_sweeps = 0;
    return 0;
}

