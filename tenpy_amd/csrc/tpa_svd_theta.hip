// The per-bond SVD section of a sweep as ONE host call (round 6; VERDICT r3 - r5: `tpa_svd_theta`).
//
// Replaces, for a two-site wave function whose bond has been decomposed before, what `svd_theta` (reference
// tenpy/linalg/truncation.py:258) -> `npc.svd` (tenpy/linalg/np_conserved.py:3676-3760, worker :4950-5002) does per charge block with
// LAPACK: the WARM route of linalg/_svd_warm.py (rounds 3 - 6) -- project on the singular vectors Bq the bond produced on its previous
// visit (W = Bq X^H), test the residual |X - W^H Bq|_F <= e_tol |X|_F per block, one-sided Jacobi on the rows of W without any QR, the
// accumulated basis Z = U'^H Bq, results into the standard layout, singular values to the host, ordered clean-up of the normalised
// vectors below the absolute floor of the stopping rule -- until round 6 about fifteen Python-driven steps with their own table plans
// (half of the steady-state calls rebuilt and uploaded those plans with the device idle).  Here every table is built in C++ per call
// (microseconds), staged in pinned memory and uploaded in ONE copy per stage; the kernels are the library's own entry points.
//
// Real data only (complex callers keep the Python route).  The stale-basis (sketch) and cold routes stay where they are: a return
// value of 1 ("residual test failed", info[0] = worst relative residual) sends the caller there, as before.
#include "tpa_common.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

constexpr int NTR = 256;
constexpr int64_t CHUNK = 16384;        // elements per workgroup of the residual reduction
constexpr int MAXD = TPA_COPY_MAXDIM;
constexpr int ORDERED_PANEL = 128;      // row panel of the triangular products of the ordered clean-up (= linalg/_svd_warm.py)

// partial sums of |A - P|^2 and |A|^2 over one chunk of the packed arena (deterministic: the host adds the partials in order)
__global__ __launch_bounds__(NTR) void theta_resid_kernel(const int64_t *__restrict__ chunks, int n_chunks, const double *__restrict__ A,
                                                          const double *__restrict__ P, double *__restrict__ part) {
    __shared__ double red[NTR / 64];
    const int c = blockIdx.x;
    const int64_t off = chunks[2 * c], len = chunks[2 * c + 1];
    double e = 0.0, a2 = 0.0;
    for (int64_t i = threadIdx.x; i < len; i += NTR) {
        const double a = A[off + i], d = a - P[off + i];
        e = fma(d, d, e);
        a2 = fma(a, a, a2);
    }
    e = block_sum<NTR>(e, red);
    a2 = block_sum<NTR>(a2, red);
    if (threadIdx.x == 0) {
        part[c] = e;
        part[n_chunks + c] = a2;
    }
}

// ---- grow-only device work area and pinned staging of this entry point (one call at a time: the mutex below) ----------------------
struct Buffers {
    char *dev = nullptr;
    size_t dev_bytes = 0;
    char *pin = nullptr;
    size_t pin_bytes = 0;
};
Buffers g_buf;
std::mutex g_mutex;

int ensure_dev(size_t bytes, hipStream_t st) {
    if (bytes <= g_buf.dev_bytes) return 0;
    TPA_HIP_CHECK(hipStreamSynchronize(st));
    if (g_buf.dev) TPA_HIP_CHECK(hipFree(g_buf.dev));
    g_buf.dev = nullptr;
    g_buf.dev_bytes = 0;
    const size_t want = bytes + bytes / 2 + (1 << 20);
    TPA_HIP_CHECK(hipMalloc((void **)&g_buf.dev, want));
    g_buf.dev_bytes = want;
    return 0;
}
int ensure_pin(size_t bytes, hipStream_t st) {
    if (bytes <= g_buf.pin_bytes) return 0;
    TPA_HIP_CHECK(hipStreamSynchronize(st));
    if (g_buf.pin) TPA_HIP_CHECK(hipHostFree(g_buf.pin));
    g_buf.pin = nullptr;
    g_buf.pin_bytes = 0;
    const size_t want = bytes + bytes / 2 + (4 << 20);
    TPA_HIP_CHECK(hipHostMalloc((void **)&g_buf.pin, want, hipHostMallocDefault));
    g_buf.pin_bytes = want;
    return 0;
}
inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- host-built tables, appended to one int64 image that goes up in one copy ------------------------------------------------------
struct GemmSpec {
    int64_t c_off, m, n, ldc, a_off, a_rs, a_ks, b_off, b_ks, b_ns, k;
};
struct GemmTab {
    size_t tasks = 0, links = 0, tiles = 0;      // int64 offsets into the image
    int n_tiles = 0;
};
struct CopyTab {
    size_t jobs = 0;
    int n = 0;
    int64_t max_elems = 0;
};
struct Image {
    std::vector<int64_t> w;
    size_t reserve(size_t n_words) {      // 256-byte aligned start
        size_t at = (w.size() + 31) & ~(size_t)31;
        w.resize(at + n_words, 0);
        return at;
    }
};

// the tables of tpa_gemm_chain for independent products (one link each), 64 x 64 tiles, longest chains first (= _svd_warm.gemm_table)
GemmTab add_gemm(Image &img, const std::vector<GemmSpec> &specs, int bm, int bn) {
    GemmTab t;
    std::vector<const GemmSpec *> use;
    for (const GemmSpec &s : specs)
        if (s.m > 0 && s.n > 0) use.push_back(&s);
    const size_t n = use.size();
    if (n == 0) return t;
    t.tasks = img.reserve(8 * n);
    t.links = img.reserve(8 * n);
    struct Tile {
        int task, row, col;
        int64_t k;
    };
    std::vector<Tile> tiles;
    for (size_t i = 0; i < n; ++i) {
        const GemmSpec &s = *use[i];
        int64_t *tk = &img.w[t.tasks + 8 * i], *lk = &img.w[t.links + 8 * i];
        tk[0] = s.c_off, tk[1] = s.m, tk[2] = s.n, tk[3] = s.ldc, tk[4] = (int64_t)i, tk[5] = 1;
        lk[0] = s.a_off, lk[1] = s.b_off, lk[2] = s.k, lk[3] = s.a_rs, lk[4] = s.a_ks, lk[5] = s.b_ks, lk[6] = s.b_ns;
        const int tm = (int)((s.m + bm - 1) / bm), tn = (int)((s.n + bn - 1) / bn);
        for (int r = 0; r < tm; ++r)
            for (int c = 0; c < tn; ++c) tiles.push_back(Tile{(int)i, r, c, s.k});
    }
    std::stable_sort(tiles.begin(), tiles.end(), [](const Tile &a, const Tile &b) { return a.k > b.k; });
    t.n_tiles = (int)tiles.size();
    t.tiles = img.reserve(2 * tiles.size());
    int32_t *tw = reinterpret_cast<int32_t *>(&img.w[t.tiles]);
    for (size_t i = 0; i < tiles.size(); ++i) {
        tw[4 * i] = tiles[i].task;
        tw[4 * i + 1] = tiles[i].row;
        tw[4 * i + 2] = tiles[i].col;
        tw[4 * i + 3] = 0;
    }
    return t;
}

struct Copy2d {
    int64_t dst_off, dst_rs, dst_cs, src_off, src_rs, src_cs, rows, cols;
};
CopyTab add_copy(Image &img, const std::vector<Copy2d> &jobs) {
    CopyTab t;
    std::vector<const Copy2d *> use;
    for (const Copy2d &j : jobs)
        if (j.rows > 0 && j.cols > 0) use.push_back(&j);
    if (use.empty()) return t;
    const int W = 4 + 3 * MAXD;
    t.n = (int)use.size();
    t.jobs = img.reserve((size_t)W * use.size());
    for (size_t i = 0; i < use.size(); ++i) {
        const Copy2d &j = *use[i];
        int64_t *r = &img.w[t.jobs + W * i];
        r[0] = j.dst_off, r[1] = j.src_off, r[2] = 2, r[3] = 0;
        r[4] = j.rows, r[5] = j.cols;
        r[4 + MAXD] = j.dst_rs, r[5 + MAXD] = j.dst_cs;
        r[4 + 2 * MAXD] = j.src_rs, r[5 + 2 * MAXD] = j.src_cs;
        t.max_elems = std::max(t.max_elems, j.rows * j.cols);
    }
    return t;
}

int run_gemm(const GemmTab &t, const int64_t *tab_dev, const void *A, const void *B, void *C, hipStream_t st) {
    if (t.n_tiles == 0) return 0;
    return tpa_gemm_chain(TPA_F64, 1, tab_dev + t.tasks, tab_dev + t.links, reinterpret_cast<const int32_t *>(tab_dev + t.tiles), t.n_tiles, A, B,
                          C, st);
}
int run_copy(const CopyTab &t, const int64_t *tab_dev, const void *src, void *dst, hipStream_t st) {
    if (t.n == 0) return 0;
    return tpa_copy_batch(TPA_F64, tab_dev + t.jobs, t.n, t.max_elems, src, dst, st);
}

#define TPA_RC(expr)          \
    do {                      \
        int _rc = (expr);     \
        if (_rc) return _rc;  \
    } while (0)

struct Blk {
    int64_t a_off, m, n, u_off, s_off, v_off, b_off, kq;
    int64_t p, l, kk;      // rows of X, length of the basis vectors, min(m, n)
    int64_t w_off, ju_off, js_off, jv_off, z_off;
};

// count of singular values above rel * |S_b|_2 (they are sorted descending inside a block; = np_conserved._svd_sig_counts)
int64_t sig_count(const double *s, int64_t k, double rel) {
    double fro2 = 0.0;
    for (int64_t i = 0; i < k; ++i) fro2 += s[i] * s[i];
    const double fro = std::sqrt(fro2);
    if (!(fro > 0.0)) return 0;
    int64_t c = 0;
    for (int64_t i = 0; i < k; ++i) c += (s[i] > rel * fro) ? 1 : 0;
    return c;
}

}  // namespace

extern "C" int tpa_svd_theta(int dtype, int side, const int64_t *blocks, int n_blocks, int64_t a_numel, const void *a_arena,
                             const void *basis_arena, void *u_arena, int64_t u_numel, void *v_arena, int64_t v_numel, double *s_host,
                             double e_tol, int lowdin_basis, int clean_iterations, double clean_floor, int alg_warm, int alg_restore,
                             int max_sweeps, double tol, int *sweeps_done, double *info, void *stream) {
    TPA_ARG_CHECK(dtype == TPA_F64);
    TPA_ARG_CHECK(side == 0 || side == 1);      // 0: 'R' (X = A, the basis spans the row space of theta), 1: 'L' (X = A^T)
    TPA_ARG_CHECK(n_blocks > 0 && n_blocks <= 65535 && blocks && s_host && info && sweeps_done);
    std::lock_guard<std::mutex> lock(g_mutex);
    hipStream_t st = (hipStream_t)stream;
    const bool R = side == 0;
    int bm = 64, bn = 64;
    tpa_gemm_tile_shape(TPA_F64, 1, &bm, &bn);
    info[0] = info[1] = info[2] = info[3] = 0.0;
    *sweeps_done = 0;

    // ---- layout -------------------------------------------------------------------------------------------------------------------
    std::vector<Blk> B(n_blocks);
    int64_t nW = 0, nJU = 0, nJS = 0, nZ = 0, expect = 0, nS = 0;
    for (int b = 0; b < n_blocks; ++b) {
        const int64_t *r = blocks + 8 * b;
        Blk &k = B[b];
        k.a_off = r[0], k.m = r[1], k.n = r[2], k.u_off = r[3], k.s_off = r[4], k.v_off = r[5], k.b_off = r[6], k.kq = r[7];
        k.p = R ? k.m : k.n;
        k.l = R ? k.n : k.m;
        k.kk = std::min(k.m, k.n);
        TPA_ARG_CHECK(k.m > 0 && k.n > 0 && k.kq > 0 && k.kq <= k.kk);
        TPA_ARG_CHECK(k.a_off == expect);      // blocks packed back to back (the residual is formed on the flat arena)
        expect += k.m * k.n;
        k.w_off = nW, k.jv_off = nW, nW += k.kq * k.p;
        k.ju_off = nJU, nJU += k.kq * k.kq;
        k.js_off = nJS, nJS += k.kq;
        k.z_off = nZ, nZ += k.kq * k.l;
        nS = std::max(nS, k.s_off + k.kk);
    }
    TPA_ARG_CHECK(expect == a_numel);
    // chunks of the residual reduction
    std::vector<int64_t> chunk_blk;
    Image img;
    std::vector<int64_t> chunks;
    for (int b = 0; b < n_blocks; ++b)
        for (int64_t o = 0; o < B[b].m * B[b].n; o += CHUNK) {
            chunks.push_back(B[b].a_off + o);
            chunks.push_back(std::min(CHUNK, B[b].m * B[b].n - o));
            chunk_blk.push_back(b);
        }
    const int n_chunks = (int)chunk_blk.size();
    // ---- tables of stages A and B ---------------------------------------------------------------------------------------------------
    std::vector<GemmSpec> sW, sP, sZ, sLG, sLM;
    std::vector<Copy2d> c1, c2;
    std::vector<int64_t> jjobs((size_t)8 * n_blocks, 0);
    for (int b = 0; b < n_blocks; ++b) {
        const Blk &k = B[b];
        const int64_t x_rs = R ? k.n : 1, x_cs = R ? 1 : k.n;      // X(j, l) = A[a_off + j x_rs + l x_cs]
        // W = Bq X^T (kq x p):  A-operand(i, t) = Bq[b_off + i l + t];  B-operand(t, j) = X(j, t)
        sW.push_back(GemmSpec{k.w_off, k.kq, k.p, k.p, k.b_off, k.l, 1, k.a_off, x_cs, x_rs, k.l});
        if (R)      // P = W^T Bq (m x n)
            sP.push_back(GemmSpec{k.a_off, k.m, k.n, k.n, k.w_off, 1, k.p, k.b_off, k.l, 1, k.kq});
        else        // P = Bq^T W (m x n)
            sP.push_back(GemmSpec{k.a_off, k.m, k.n, k.n, k.b_off, 1, k.l, k.w_off, k.p, 1, k.kq});
        int64_t *j = &jjobs[(size_t)8 * b];
        j[0] = k.w_off, j[1] = k.kq, j[2] = k.p, j[3] = k.ju_off, j[4] = k.js_off, j[5] = k.jv_off, j[6] = 1;      // rows of W
        // Z = U'^T Bq (kq x l):  A-operand(i, t) = JU[ju_off + i + t kq]
        sZ.push_back(GemmSpec{k.z_off, k.kq, k.l, k.l, k.ju_off, 1, k.kq, k.b_off, k.l, 1, k.kq});
        // one first-order Loewdin step on the rows of Z (every 8th warm generation of a bond): G = Z Z^T, T2 = G Z
        sLG.push_back(GemmSpec{k.ju_off, k.kq, k.kq, k.kq, k.z_off, k.l, 1, k.z_off, 1, k.l, k.l});
        sLM.push_back(GemmSpec{k.z_off, k.kq, k.l, k.l, k.ju_off, k.kq, 1, k.z_off, k.l, 1, k.kq});
        if (R) {    // VH_A rows = Z;  U_A[j][i] = VH'[i][j]
            c1.push_back(Copy2d{k.v_off, k.n, 1, k.z_off, k.l, 1, k.kq, k.n});
            c2.push_back(Copy2d{k.u_off, k.kk, 1, k.jv_off, 1, k.p, k.m, k.kq});
        } else {    // U_A[l][i] = Z[i][l];  VH_A = VH'
            c1.push_back(Copy2d{k.u_off, k.kk, 1, k.z_off, 1, k.l, k.m, k.kq});
            c2.push_back(Copy2d{k.v_off, k.n, 1, k.jv_off, k.p, 1, k.kq, k.n});
        }
    }
    const GemmTab tW = add_gemm(img, sW, bm, bn), tP = add_gemm(img, sP, bm, bn), tZ = add_gemm(img, sZ, bm, bn);
    const GemmTab tLG = lowdin_basis ? add_gemm(img, sLG, bm, bn) : GemmTab(), tLM = lowdin_basis ? add_gemm(img, sLM, bm, bn) : GemmTab();
    const CopyTab tC1 = add_copy(img, c1), tC2 = add_copy(img, c2);
    const size_t o_chunks = img.reserve(chunks.size());
    std::memcpy(&img.w[o_chunks], chunks.data(), chunks.size() * 8);
    const size_t tabA_words = img.w.size();

    // ---- device work area ---------------------------------------------------------------------------------------------------------
    const int64_t svd_work = tpa_svd_worksize(TPA_F64, jjobs.data(), n_blocks);
    TPA_ARG_CHECK(svd_work >= 0);
    int64_t max_len = 0, sum_kk_len = 0, sum_kk2 = 0;
    for (const Blk &k : B) {
        const int64_t len = R ? k.m : k.n;      // length of the vectors the clean-up treats (columns of U: m; rows of VH: n)
        max_len = std::max(max_len, len);
        sum_kk_len += k.kk * len;
        sum_kk2 += k.kk * k.kk;
    }
    // upper bound of the clean-up tables (built after the singular values are known): per block two copy jobs, one triangle job and
    // per 128-row panel two GEMM tasks with their tiles
    size_t cap_words = 512;
    for (const Blk &k : B) {
        const int64_t len = R ? k.m : k.n, npan = (k.kk + ORDERED_PANEL - 1) / ORDERED_PANEL;
        cap_words += 2 * (4 + 3 * MAXD) + 2 + (size_t)npan * 32;
        for (int64_t r0 = 0; r0 < k.kk; r0 += ORDERED_PANEL) {
            const int64_t r1 = std::min<int64_t>(r0 + ORDERED_PANEL, k.kk);
            const int64_t tr = (r1 - r0 + bm - 1) / bm;
            cap_words += 2 * (size_t)(tr * ((r1 + bn - 1) / bn) + tr * ((len + bn - 1) / bn));
        }
    }
    const size_t tab_cap = al(8 * cap_words);
    size_t o = 0;
    const size_t o_tabA = o;
    o += al(8 * tabA_words);
    const size_t o_tabC = o;      // tables of the clean-up (second upload)
    o += tab_cap;
    const size_t o_W = o;
    o += al(8 * (size_t)nW);
    const size_t o_P = o;
    o += al(8 * (size_t)a_numel);
    const size_t o_part = o;
    o += al(16 * (size_t)n_chunks);
    const size_t o_JU = o;
    o += al(8 * (size_t)nJU);
    const size_t o_JS = o;
    o += al(8 * (size_t)nJS);
    const size_t o_JV = o;
    o += al(8 * (size_t)nW);
    const size_t o_Z = o;
    o += al(8 * (size_t)nZ);
    const size_t o_T = o;          // clean-up: gathered vectors, their Gram matrices, the products
    o += al(8 * (size_t)std::max<int64_t>(sum_kk_len, nZ));
    const size_t o_T2 = o;
    o += al(8 * (size_t)std::max<int64_t>(sum_kk_len, nZ));
    const size_t o_G = o;
    o += al(8 * (size_t)std::max<int64_t>(sum_kk2, nJU));
    const size_t o_svd = o;
    o += al((size_t)svd_work);
    TPA_RC(ensure_dev(o, st));
    const size_t pin_need = al(8 * tabA_words) + tab_cap + al(16 * (size_t)n_chunks) + al(8 * (size_t)nJS);
    TPA_RC(ensure_pin(pin_need, st));
    char *dv = g_buf.dev, *ph = g_buf.pin;
    const size_t p_tabA = 0, p_tabC = al(8 * tabA_words), p_part = p_tabC + tab_cap, p_S = p_part + al(16 * (size_t)n_chunks);
    double *W = (double *)(dv + o_W), *P = (double *)(dv + o_P), *part = (double *)(dv + o_part), *JU = (double *)(dv + o_JU),
           *JS = (double *)(dv + o_JS), *JV = (double *)(dv + o_JV), *Z = (double *)(dv + o_Z), *T = (double *)(dv + o_T),
           *T2 = (double *)(dv + o_T2), *G = (double *)(dv + o_G);
    const int64_t *tabA = (const int64_t *)(dv + o_tabA);

    // ---- stage A: W, the part of A the basis spans, the residual -------------------------------------------------------------------
    std::memcpy(ph + p_tabA, img.w.data(), 8 * tabA_words);
    TPA_HIP_CHECK(hipMemcpyAsync(dv + o_tabA, ph + p_tabA, 8 * tabA_words, hipMemcpyHostToDevice, st));
    TPA_RC(run_gemm(tW, tabA, basis_arena, a_arena, W, st));
    if (R)
        TPA_RC(run_gemm(tP, tabA, W, basis_arena, P, st));
    else
        TPA_RC(run_gemm(tP, tabA, basis_arena, W, P, st));
    theta_resid_kernel<<<n_chunks, NTR, 0, st>>>(tabA + o_chunks, n_chunks, (const double *)a_arena, P, part);
    TPA_LAUNCH_CHECK();
    TPA_HIP_CHECK(hipMemcpyAsync(ph + p_part, part, 16 * (size_t)n_chunks, hipMemcpyDeviceToHost, st));
    // (the result arenas are cleared while the host waits for the test)
    TPA_HIP_CHECK(hipMemsetAsync(u_arena, 0, 8 * (size_t)u_numel, st));
    TPA_HIP_CHECK(hipMemsetAsync(v_arena, 0, 8 * (size_t)v_numel, st));
    TPA_HIP_CHECK(hipStreamSynchronize(st));
    {
        const double *pp = (const double *)(ph + p_part);
        std::vector<double> e2(n_blocks, 0.0), a2(n_blocks, 0.0);
        for (int c = 0; c < n_chunks; ++c) {
            e2[chunk_blk[c]] += pp[c];
            a2[chunk_blk[c]] += pp[n_chunks + c];
        }
        double worst = 0.0;
        int n_stale = 0;
        for (int b = 0; b < n_blocks; ++b) {
            const bool ok = std::isfinite(e2[b]) && std::isfinite(a2[b]) && a2[b] > 0.0;
            const double e_rel = ok ? std::sqrt(e2[b] / a2[b]) : INFINITY;
            worst = std::max(worst, std::isfinite(e_rel) ? e_rel : 1.0);
            n_stale += (e_rel <= e_tol) ? 0 : 1;
        }
        info[0] = worst;
        info[1] = (double)n_stale;
        if (n_stale) return 1;
    }

    // ---- stage B: Jacobi on the rows of W, accumulated basis, results ---------------------------------------------------------------
    {
        if (alg_warm != alg_restore) tpa_svd_set_algorithm(alg_warm);
        const int rc = tpa_svd_batch(TPA_F64, jjobs.data(), n_blocks, W, JU, JS, JV, dv + o_svd, svd_work, max_sweeps, tol, sweeps_done, st);
        if (alg_warm != alg_restore) tpa_svd_set_algorithm(alg_restore);
        if (rc) return rc;
    }
    TPA_HIP_CHECK(hipMemcpyAsync(ph + p_S, JS, 8 * (size_t)nJS, hipMemcpyDeviceToHost, st));
    // the host waits for the singular values only (event right behind their copy): the accumulated basis, the result copies and the
    // clean-up below run while the caller already works on the truncation
    static hipEvent_t ev_S = nullptr;
    if (!ev_S) TPA_HIP_CHECK(hipEventCreateWithFlags(&ev_S, hipEventDisableTiming));
    TPA_HIP_CHECK(hipEventRecord(ev_S, st));
    TPA_RC(run_gemm(tZ, tabA, JU, basis_arena, Z, st));
    if (lowdin_basis) {      // Z <- (3 Z - (Z Z^T) Z) / 2   (G lives in the JU area: U' has been consumed)
        TPA_RC(run_gemm(tLG, tabA, Z, Z, JU, st));
        TPA_RC(run_gemm(tLM, tabA, JU, Z, T2, st));      // (T2 laid out like Z)
        TPA_RC(tpa_scal(TPA_F64, nZ, 1.5, 0.0, Z, st));
        TPA_RC(tpa_axpy(TPA_F64, nZ, -0.5, 0.0, T2, Z, st));
    }
    TPA_RC(run_copy(tC1, tabA, Z, R ? v_arena : u_arena, st));
    TPA_RC(run_copy(tC2, tabA, JV, R ? u_arena : v_arena, st));
    TPA_HIP_CHECK(hipEventSynchronize(ev_S));
    const double *SJ = (const double *)(ph + p_S);
    for (int64_t i = 0; i < nS; ++i) s_host[i] = 0.0;
    for (const Blk &k : B)
        for (int64_t i = 0; i < k.kq; ++i) {
            const double v = SJ[k.js_off + i];
            if (std::isnan(v)) {
                snprintf(tpa_errbuf, sizeof(tpa_errbuf), "tpa_svd_theta: NaN in the singular values");
                return TPA_E_NAN;
            }
            s_host[k.s_off + i] = v;
        }

    // ---- ordered clean-up of the normalised Jacobi rows (columns of U for side R, rows of VH for side L; = _svd_clean_small) -------------
    if (clean_iterations > 0 && clean_floor > 0.0) {
        struct Cl {
            int64_t off, nv, len, vs, cs, t_off, g_off;
        };
        std::vector<Cl> cl;
        int64_t nT = 0, nG = 0, max_g = 0;
        for (const Blk &k : B) {
            const int64_t k0 = sig_count(s_host + k.s_off, k.kk, clean_floor), k1 = sig_count(s_host + k.s_off, k.kk, 1e-15);
            if (k1 - k0 <= 0) continue;
            const int64_t nv = std::min((k1 + 31) / 32 * 32, k.kk);
            if (nv <= 1) continue;
            Cl c;
            c.nv = nv;
            if (R) c.off = k.u_off, c.len = k.m, c.vs = 1, c.cs = k.kk;
            else c.off = k.v_off, c.len = k.n, c.vs = k.n, c.cs = 1;
            c.t_off = nT, nT += nv * c.len;
            c.g_off = nG, nG += nv * nv;
            max_g = std::max(max_g, nv * nv);
            cl.push_back(c);
        }
        if (!cl.empty()) {
            Image ic;
            std::vector<Copy2d> gat, sca;
            std::vector<GemmSpec> gram, mult;
            std::vector<int64_t> tri;
            for (const Cl &c : cl) {
                gat.push_back(Copy2d{c.t_off, c.len, 1, c.off, c.vs, c.cs, c.nv, c.len});
                sca.push_back(Copy2d{c.off, c.vs, c.cs, c.t_off, c.len, 1, c.nv, c.len});
                tri.push_back(c.g_off);
                tri.push_back(c.nv);
                for (int64_t r0 = 0; r0 < c.nv; r0 += ORDERED_PANEL) {
                    const int64_t r1 = std::min(r0 + ORDERED_PANEL, c.nv);
                    // G[r0:r1, 0:r1] = T[r0:r1] T[0:r1]^T;  T2[r0:r1] = N[r0:r1, 0:r1] T[0:r1]
                    gram.push_back(GemmSpec{c.g_off + r0 * c.nv, r1 - r0, r1, c.nv, c.t_off + r0 * c.len, c.len, 1, c.t_off, 1, c.len, c.len});
                    mult.push_back(GemmSpec{c.t_off + r0 * c.len, r1 - r0, c.len, c.len, c.g_off + r0 * c.nv, c.nv, 1, c.t_off, c.len, 1, r1});
                }
            }
            const CopyTab tG = add_copy(ic, gat), tS = add_copy(ic, sca);
            const GemmTab tGr = add_gemm(ic, gram, bm, bn), tMu = add_gemm(ic, mult, bm, bn);
            const size_t o_tri = ic.reserve(tri.size());
            std::memcpy(&ic.w[o_tri], tri.data(), tri.size() * 8);
            if (8 * ic.w.size() > tab_cap) {
                snprintf(tpa_errbuf, sizeof(tpa_errbuf), "tpa_svd_theta: clean-up tables (%zu bytes) exceed their area (%zu)", 8 * ic.w.size(), tab_cap);
                return TPA_E_NOMEM;
            }
            std::memcpy(ph + p_tabC, ic.w.data(), 8 * ic.w.size());
            TPA_HIP_CHECK(hipMemcpyAsync(dv + o_tabC, ph + p_tabC, 8 * ic.w.size(), hipMemcpyHostToDevice, st));
            const int64_t *tabC = (const int64_t *)(dv + o_tabC);
            void *arena = R ? u_arena : v_arena;
            TPA_RC(run_copy(tG, tabC, arena, T, st));
            for (int it = 0; it < clean_iterations; ++it) {
                TPA_RC(run_gemm(tGr, tabC, T, T, G, st));
                TPA_RC(tpa_tri_lower_batch(TPA_F64, tabC + o_tri, (int)cl.size(), max_g, G, st));
                TPA_RC(run_gemm(tMu, tabC, G, T, T2, st));
                TPA_RC(tpa_axpy(TPA_F64, nT, -1.0, 0.0, T2, T, st));
            }
            TPA_RC(run_copy(tS, tabC, T, arena, st));
            info[2] = (double)cl.size();
        }
    }
    return 0;
}

// Remember the significant singular vectors of a decomposition as the bond's next warm-start bases (= np_conserved._svd_warm_store):
// rows [0, ksig_b) of VH_b into `basis_r` (ksig x n row-major, back to back) and columns [0, ksig_b) of U_b, transposed, into `basis_l`
// (ksig x m).  blocks: int64[n][8] as for tpa_svd_theta (only m, n, u_off, v_off are read); tables built and uploaded here.
extern "C" int tpa_svd_theta_store(int dtype, const int64_t *blocks, const int64_t *ksig, int n_blocks, const void *u_arena, const void *v_arena,
                                   void *basis_r, void *basis_l, void *stream) {
    TPA_ARG_CHECK(dtype == TPA_F64 && n_blocks > 0 && n_blocks <= 65535 && blocks && ksig);
    std::lock_guard<std::mutex> lock(g_mutex);
    hipStream_t st = (hipStream_t)stream;
    std::vector<Copy2d> cr, clv;
    int64_t r_off = 0, l_off = 0;
    for (int b = 0; b < n_blocks; ++b) {
        const int64_t *r = blocks + 8 * b;
        const int64_t m = r[1], n = r[2], u_off = r[3], v_off = r[5], kk = std::min(m, n), k = ksig[b];
        TPA_ARG_CHECK(k >= 0 && k <= kk);
        cr.push_back(Copy2d{r_off, n, 1, v_off, n, 1, k, n});
        clv.push_back(Copy2d{l_off, m, 1, u_off, 1, kk, k, m});
        r_off += k * n;
        l_off += k * m;
    }
    Image img;
    const CopyTab tR = add_copy(img, cr), tL = add_copy(img, clv);
    if (img.w.empty()) return 0;
    // table buffers of their own (two slots used in turn: the copies of the previous store may still be queued when the next bond's
    // store is staged only in theory -- a whole bond update with its waits lies in between -- but a slot is never rewritten while
    // the upload out of it or the kernels reading it can be in flight)
    static char *s_dev[2] = {nullptr, nullptr}, *s_pin[2] = {nullptr, nullptr};
    static size_t s_bytes[2] = {0, 0};
    static int s_turn = 0;
    const int slot = (s_turn++) & 1;
    const size_t bytes = 8 * img.w.size();
    if (bytes > s_bytes[slot]) {
        TPA_HIP_CHECK(hipStreamSynchronize(st));
        if (s_dev[slot]) TPA_HIP_CHECK(hipFree(s_dev[slot]));
        if (s_pin[slot]) TPA_HIP_CHECK(hipHostFree(s_pin[slot]));
        s_dev[slot] = s_pin[slot] = nullptr;
        s_bytes[slot] = 0;
        const size_t want = 2 * bytes + 4096;
        TPA_HIP_CHECK(hipMalloc((void **)&s_dev[slot], want));
        TPA_HIP_CHECK(hipHostMalloc((void **)&s_pin[slot], want, hipHostMallocDefault));
        s_bytes[slot] = want;
    }
    std::memcpy(s_pin[slot], img.w.data(), bytes);
    TPA_HIP_CHECK(hipMemcpyAsync(s_dev[slot], s_pin[slot], bytes, hipMemcpyHostToDevice, st));
    const int64_t *tab = (const int64_t *)s_dev[slot];
    TPA_RC(run_copy(tR, tab, v_arena, basis_r, st));
    TPA_RC(run_copy(tL, tab, u_arena, basis_l, st));
    return 0;
}
