// K1: grouped + chained block GEMM on fp64 MFMA (v_mfma_f64_16x16x4_f64), gfx950.
//
// Replaces CblasGemmBatch.run (reference tenpy/linalg/_npc_helper.pyx:204-273): one launch executes
// every (A,B,C) triple of a block-sparse tensordot.  The reference's accumulation "levels"
// (beta=0 for level 0, beta=1 afterwards) become a *chain* per C tile that is summed in the MFMA
// accumulators, so C is written exactly once and never re-read.
//
// Work decomposition: one workgroup per BM x BN tile of one C block (host-built tile table, largest
// chains first).  4 waves (2x2); a wave owns a (TM*16) x (TN*16) sub-tile = TM x TN MFMA tiles.
// Operand tiles are staged global -> registers -> LDS (register prefetch of the next k-tile overlaps
// the MFMAs of the current one); the LDS image of an operand follows its *global* fast axis so both
// the global load and the LDS store are unit-stride, and the MFMA fragment reads adapt instead.
#include "tpa_common.h"

typedef double d4 __attribute__((ext_vector_type(4)));

namespace {

struct Link {  // int64[8]
    int64_t a_off, b_off, k, a_rs, a_ks, b_ks, b_ns, flags;
};
struct Task {  // int64[8]
    int64_t c_off, m, n, ldc, link_begin, link_count, accumulate, pad;
};

constexpr int BK = 16;

template <bool CPLX, int BM, int BN, int TM, int TN>
struct Cfg {
    static constexpr int WM = BM / (TM * 16);
    static constexpr int WN = BN / (TN * 16);
    static constexpr int NT = WM * WN * 64;
    static constexpr int PADA = 16, PADB = 16;
    // LDS doubles per plane of one operand: max over the two layouts
    static constexpr int A_LDS = (BM * (BK + 1) > BK * (BM + PADA)) ? BM * (BK + 1) : BK * (BM + PADA);
    static constexpr int B_LDS = (BN * (BK + 1) > BK * (BN + PADB)) ? BN * (BK + 1) : BK * (BN + PADB);
    static constexpr int PLANES = CPLX ? 2 : 1;
    static constexpr int EA = BM * BK / NT;  // elements of A staged per thread per k-tile
    static constexpr int EB = BN * BK / NT;
};

template <bool CPLX, int BM, int BN, int TM, int TN>
__global__ __launch_bounds__((Cfg<CPLX, BM, BN, TM, TN>::NT)) void gemm_chain_kernel(
    const Task *__restrict__ tasks, const Link *__restrict__ links, const int4 *__restrict__ tiles,
    const double *__restrict__ Abase, const double *__restrict__ Bbase, double *__restrict__ Cbase) {
    using C = Cfg<CPLX, BM, BN, TM, TN>;
    constexpr int NT = C::NT, EA = C::EA, EB = C::EB, PL = C::PLANES;
    __shared__ double lds[PL * (C::A_LDS + C::B_LDS)];
    double *As = lds;
    double *Bs = lds + PL * C::A_LDS;

    const int4 tile = tiles[blockIdx.x];
    const Task tk = tasks[tile.x];
    const int m = (int)tk.m, n = (int)tk.n;
    const int row0 = tile.y * BM, col0 = tile.z * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / C::WN, wc = wave % C::WN;
    const int l15 = lane & 15, l4 = lane >> 4;

    d4 acc[PL][TM][TN];
#pragma unroll
    for (int p = 0; p < PL; ++p)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[p][i][j] = d4{0, 0, 0, 0};

    // ---- chain iteration state: (link index, k0) --------------------------------------------
    int li = 0;
    const int nl = (int)tk.link_count;
    const Link *lk = links + tk.link_begin;
    // skip empty links
    while (li < nl && lk[li].k <= 0) ++li;
    int k0 = 0;

    double ra[PL][EA], rb[PL][EB];
    int lsa_i = 0, lsa_k = 0, lsb_k = 0, lsb_j = 0;  // LDS strides of the *staged* tile

    auto load_global = [&](int li_, int k0_, int &sa_i, int &sa_k, int &sb_k, int &sb_j) {
        const Link L = lk[li_];
        const int kk_tot = (int)L.k;
        const bool a_kfast = (L.a_ks == 1);
        const bool b_kfast = (L.b_ks == 1) && (L.b_ns != 1);
        const double sgn_a = (CPLX && (L.flags & 1)) ? -1.0 : 1.0;
        const double sgn_b = (CPLX && (L.flags & 2)) ? -1.0 : 1.0;
        sa_i = a_kfast ? (BK + 1) : 1;
        sa_k = a_kfast ? 1 : (BM + C::PADA);
        sb_j = b_kfast ? (BK + 1) : 1;
        sb_k = b_kfast ? 1 : (BN + C::PADB);
        const double *Ap = Abase + (CPLX ? 2 : 1) * L.a_off;
        const double *Bp = Bbase + (CPLX ? 2 : 1) * L.b_off;
#pragma unroll
        for (int r = 0; r < EA; ++r) {
            const int e = tid + r * NT;
            const int i = a_kfast ? (e / BK) : (e % BM);
            const int kk = a_kfast ? (e % BK) : (e / BM);
            const bool ok = (row0 + i < m) && (k0_ + kk < kk_tot);
            const int64_t g = (int64_t)(row0 + i) * L.a_rs + (int64_t)(k0_ + kk) * L.a_ks;
            if (CPLX) {
                double2 v = ok ? *reinterpret_cast<const double2 *>(Ap + 2 * g) : double2{0, 0};
                ra[0][r] = v.x;
                ra[PL - 1][r] = sgn_a * v.y;
            } else {
                ra[0][r] = ok ? Ap[g] : 0.0;
            }
        }
#pragma unroll
        for (int r = 0; r < EB; ++r) {
            const int e = tid + r * NT;
            const int j = b_kfast ? (e / BK) : (e % BN);
            const int kk = b_kfast ? (e % BK) : (e / BN);
            const bool ok = (col0 + j < n) && (k0_ + kk < kk_tot);
            const int64_t g = (int64_t)(k0_ + kk) * L.b_ks + (int64_t)(col0 + j) * L.b_ns;
            if (CPLX) {
                double2 v = ok ? *reinterpret_cast<const double2 *>(Bp + 2 * g) : double2{0, 0};
                rb[0][r] = v.x;
                rb[PL - 1][r] = sgn_b * v.y;
            } else {
                rb[0][r] = ok ? Bp[g] : 0.0;
            }
        }
    };

    auto store_lds = [&](int sa_i, int sa_k, int sb_k, int sb_j) {
        const bool a_kfast = (sa_k == 1);
        const bool b_kfast = (sb_k == 1);
#pragma unroll
        for (int r = 0; r < EA; ++r) {
            const int e = tid + r * NT;
            const int i = a_kfast ? (e / BK) : (e % BM);
            const int kk = a_kfast ? (e % BK) : (e / BM);
#pragma unroll
            for (int p = 0; p < PL; ++p) As[p * C::A_LDS + i * sa_i + kk * sa_k] = ra[p][r];
        }
#pragma unroll
        for (int r = 0; r < EB; ++r) {
            const int e = tid + r * NT;
            const int j = b_kfast ? (e / BK) : (e % BN);
            const int kk = b_kfast ? (e % BK) : (e / BN);
#pragma unroll
            for (int p = 0; p < PL; ++p) Bs[p * C::B_LDS + kk * sb_k + j * sb_j] = rb[p][r];
        }
    };

    bool have = (li < nl);
    if (have) load_global(li, k0, lsa_i, lsa_k, lsb_k, lsb_j);

    while (have) {
        // stage the prefetched registers
        const int ca_i = lsa_i, ca_k = lsa_k, cb_k = lsb_k, cb_j = lsb_j;
        store_lds(ca_i, ca_k, cb_k, cb_j);
        __syncthreads();
        // advance and prefetch next k-tile into registers
        k0 += BK;
        if (k0 >= (int)lk[li].k) {
            k0 = 0;
            ++li;
            while (li < nl && lk[li].k <= 0) ++li;
        }
        have = (li < nl);
        if (have) load_global(li, k0, lsa_i, lsa_k, lsb_k, lsb_j);

        // MFMA on the staged tile
        const double *Aw = As + (wr * TM * 16 + l15) * ca_i + l4 * ca_k;
        const double *Bw = Bs + (wc * TN * 16 + l15) * cb_j + l4 * cb_k;
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            double a[PL][TM], b[PL][TN];
#pragma unroll
            for (int p = 0; p < PL; ++p) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[p][i] = Aw[p * C::A_LDS + i * 16 * ca_i + ks * 4 * ca_k];
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[p][j] = Bw[p * C::B_LDS + j * 16 * cb_j + ks * 4 * cb_k];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (CPLX) {
                        acc[0][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0][i], b[0][j], acc[0][i][j], 0, 0, 0);
                        acc[0][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[PL - 1][i], b[PL - 1][j], acc[0][i][j], 0, 0, 0);
                        acc[PL - 1][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0][i], b[PL - 1][j], acc[PL - 1][i][j], 0, 0, 0);
                        acc[PL - 1][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[PL - 1][i], b[0][j], acc[PL - 1][i][j], 0, 0, 0);
                    } else {
                        acc[0][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0][i], b[0][j], acc[0][i][j], 0, 0, 0);
                    }
                }
        }
        __syncthreads();
    }

    // ---- epilogue: C/D layout of v_mfma_f64_16x16x4_f64: col = lane&15, row = (lane>>4) + 4*reg
    double *Cp = Cbase + (CPLX ? 2 : 1) * tk.c_off;
    const bool accum = tk.accumulate != 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + wr * TM * 16 + i * 16 + l4 + 4 * r;
                const int col = col0 + wc * TN * 16 + j * 16 + l15;
                if (row < m && col < n) {
                    const int64_t g = (int64_t)row * tk.ldc + col;
                    if (CPLX) {
                        double2 v{acc[0][i][j][r], acc[PL - 1][i][j][r]};
                        double2 *dst = reinterpret_cast<double2 *>(Cp + 2 * g);
                        if (accum) {
                            double2 o = *dst;
                            v.x += o.x;
                            v.y += o.y;
                        }
                        *dst = v;
                    } else {
                        double v = acc[0][i][j][r];
                        if (accum) v += Cp[g];
                        Cp[g] = v;
                    }
                }
            }
}

}  // namespace

extern "C" int tpa_gemm_tile_shape(int dtype, int *bm, int *bn) {
    if (dtype == TPA_F64) {
        *bm = 128;
        *bn = 128;
    } else {
        *bm = 128;
        *bn = 64;
    }
    return 0;
}

extern "C" int tpa_gemm_chain(int dtype, const int64_t *tasks_dev, const int64_t *links_dev,
                              const int32_t *tiles_dev, int n_tiles, const void *Abase,
                              const void *Bbase, void *Cbase, void *stream) {
    TPA_ARG_CHECK(dtype == TPA_F64 || dtype == TPA_C128);
    if (n_tiles <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == TPA_F64) {
        using C = Cfg<false, 128, 128, 4, 4>;
        gemm_chain_kernel<false, 128, 128, 4, 4><<<n_tiles, C::NT, 0, st>>>(
            (const Task *)tasks_dev, (const Link *)links_dev, (const int4 *)tiles_dev,
            (const double *)Abase, (const double *)Bbase, (double *)Cbase);
    } else {
        using C = Cfg<true, 128, 64, 4, 2>;
        gemm_chain_kernel<true, 128, 64, 4, 2><<<n_tiles, C::NT, 0, st>>>(
            (const Task *)tasks_dev, (const Link *)links_dev, (const int4 *)tiles_dev,
            (const double *)Abase, (const double *)Bbase, (double *)Cbase);
    }
    TPA_LAUNCH_CHECK();
    return 0;
}
