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

template <bool CPLX, int BM, int BN, int TM, int TN, int BK_>
struct Cfg {
    static constexpr int BK = BK_;
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
    static_assert(NT % BK == 0 && NT % BM == 0 && NT % BN == 0, "staging offsets must be affine in the element index");
};

// What a thread needs to stage its share of one k-tile of one link: element r of the thread sits at
// (global) base + r * gstep (+ k0 * kstep), (LDS) lbase + r * lstep; its k index inside the tile is kk0 + r * kkstep.
// NT is a multiple of BK, BM and BN, so these are affine in r -- no per-element address arrays.
struct Stage {
    const double *p;      // advances by kstep per k-tile
    int64_t gstep, kstep;
    int lbase, lstep, kk0, kkstep;
    int i0, istep, lim;   // row (A) / column (B) index of element r = i0 + r * istep, valid while < lim
};

// DB (round 5): two LDS images of the operand tiles.  The stores of k-tile t + 1 go to the image the MFMAs are NOT reading, so one
// barrier per k-tile is left (the single-image loop needs two: stores -> reads, reads -> next stores) and the LDS stores overlap the
// matrix pipe instead of preceding it.  40 KB for the real 64 x 64 tile: four workgroups per CU, as the registers allow anyway.
template <bool CPLX, int BM, int BN, int TM, int TN, int BK_, bool DB>
__global__ __launch_bounds__((Cfg<CPLX, BM, BN, TM, TN, BK_>::NT)) void gemm_chain_kernel(
    const Task *__restrict__ tasks, const Link *__restrict__ links, const int4 *__restrict__ tiles,
    const double *__restrict__ Abase, const double *__restrict__ Bbase, double *__restrict__ Cbase) {
    using C = Cfg<CPLX, BM, BN, TM, TN, BK_>;
    constexpr int NT = C::NT, EA = C::EA, EB = C::EB, PL = C::PLANES, BK = C::BK;
    constexpr int ES = CPLX ? 2 : 1;  // doubles per element
    constexpr int IMG = PL * (C::A_LDS + C::B_LDS);      // doubles per LDS image
    __shared__ double lds[(DB ? 2 : 1) * IMG];
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

    const int nl = (int)tk.link_count;
    const Link *lk = links + tk.link_begin;

    // ---- link state: `cur` is being multiplied out of LDS, `nxt` is what the register prefetch belongs to (the next k-tile
    //      of the same link or the FIRST k-tile of the next link: the chain never drains its load pipeline at a link boundary)
    struct LinkView {
        Stage a, b;
        int K, sa_i, sa_k, sb_j, sb_k;
        double sgn_a, sgn_b;
        int64_t oa[EA], ob[EB];      // element offsets (doubles) relative to a.p / b.p, fixed per link
    };
    auto open_link = [&](int li, LinkView &v) {
        const Link L = lk[li];
        v.K = (int)L.k;
        const bool a_kfast = (L.a_ks == 1);
        const bool b_kfast = (L.b_ks == 1) && (L.b_ns != 1);
        v.sgn_a = (CPLX && (L.flags & 1)) ? -1.0 : 1.0;
        v.sgn_b = (CPLX && (L.flags & 2)) ? -1.0 : 1.0;
        v.sa_i = a_kfast ? (BK + 1) : 1;
        v.sa_k = a_kfast ? 1 : (BM + C::PADA);
        v.sb_j = b_kfast ? (BK + 1) : 1;
        v.sb_k = b_kfast ? 1 : (BN + C::PADB);
        {   // A: element e = tid + r NT;  k-fast: (i, kk) = (e / BK, e % BK);  else (e % BM, e / BM)
            const int i = a_kfast ? (tid / BK) : (tid % BM), kk = a_kfast ? (tid % BK) : (tid / BM);
            const int di = a_kfast ? (NT / BK) : 0, dk = a_kfast ? 0 : (NT / BM);
            v.a.p = Abase + ES * (L.a_off + (int64_t)(row0 + i) * L.a_rs + (int64_t)kk * L.a_ks);
            v.a.gstep = ES * ((int64_t)di * L.a_rs + (int64_t)dk * L.a_ks);
            v.a.kstep = ES * (int64_t)BK * L.a_ks;
            v.a.lbase = i * v.sa_i + kk * v.sa_k;
            v.a.lstep = di * v.sa_i + dk * v.sa_k;
            v.a.kk0 = kk;
            v.a.kkstep = dk;
            v.a.i0 = row0 + i;
            v.a.istep = di;
            v.a.lim = m;
#pragma unroll
            for (int r = 0; r < EA; ++r) v.oa[r] = r * v.a.gstep;
        }
        {
            const int j = b_kfast ? (tid / BK) : (tid % BN), kk = b_kfast ? (tid % BK) : (tid / BN);
            const int dj = b_kfast ? (NT / BK) : 0, dk = b_kfast ? 0 : (NT / BN);
            v.b.p = Bbase + ES * (L.b_off + (int64_t)kk * L.b_ks + (int64_t)(col0 + j) * L.b_ns);
            v.b.gstep = ES * ((int64_t)dk * L.b_ks + (int64_t)dj * L.b_ns);
            v.b.kstep = ES * (int64_t)BK * L.b_ks;
            v.b.lbase = kk * v.sb_k + j * v.sb_j;
            v.b.lstep = dk * v.sb_k + dj * v.sb_j;
            v.b.kk0 = kk;
            v.b.kkstep = dk;
            v.b.i0 = col0 + j;
            v.b.istep = dj;
            v.b.lim = n;
#pragma unroll
            for (int r = 0; r < EB; ++r) v.ob[r] = r * v.b.gstep;
        }
    };
    double ra[PL][EA], rb[PL][EB];
    auto load_tile = [&](LinkView &v, int k0) {      // registers <- k-tile k0 of link v; advances the link's pointers
        const int krem = v.K - k0;  // >= 1
        const bool full = (krem >= BK);
#pragma unroll
        for (int r = 0; r < EA; ++r) {
            const bool ok = (v.a.i0 + r * v.a.istep < v.a.lim) && (full || v.a.kk0 + r * v.a.kkstep < krem);
            const double *q = v.a.p + v.oa[r];
            if (CPLX) {
                double2 x = ok ? *reinterpret_cast<const double2 *>(q) : double2{0, 0};
                ra[0][r] = x.x;
                ra[PL - 1][r] = v.sgn_a * x.y;
            } else {
                ra[0][r] = ok ? *q : 0.0;
            }
        }
#pragma unroll
        for (int r = 0; r < EB; ++r) {
            const bool ok = (v.b.i0 + r * v.b.istep < v.b.lim) && (full || v.b.kk0 + r * v.b.kkstep < krem);
            const double *q = v.b.p + v.ob[r];
            if (CPLX) {
                double2 x = ok ? *reinterpret_cast<const double2 *>(q) : double2{0, 0};
                rb[0][r] = x.x;
                rb[PL - 1][r] = v.sgn_b * x.y;
            } else {
                rb[0][r] = ok ? *q : 0.0;
            }
        }
        v.a.p += v.a.kstep;
        v.b.p += v.b.kstep;
    };

    auto store_tile = [&](const LinkView &v, double *Ad, double *Bd) {      // registers -> one LDS image, in link v's layout
#pragma unroll
        for (int r = 0; r < EA; ++r)
#pragma unroll
            for (int p = 0; p < PL; ++p) Ad[p * C::A_LDS + v.a.lbase + r * v.a.lstep] = ra[p][r];
#pragma unroll
        for (int r = 0; r < EB; ++r)
#pragma unroll
            for (int p = 0; p < PL; ++p) Bd[p * C::B_LDS + v.b.lbase + r * v.b.lstep] = rb[p][r];
    };
    auto mma_tile = [&](const double *Aw, const double *Bw, int sa_i, int sa_k, int sb_j, int sb_k) {
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            double a[PL][TM], b[PL][TN];
#pragma unroll
            for (int p = 0; p < PL; ++p) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[p][i] = Aw[p * C::A_LDS + i * 16 * sa_i + ks * 4 * sa_k];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[p][j] = Bw[p * C::B_LDS + j * 16 * sb_j + ks * 4 * sb_k];
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
    };

    int li = 0;
    while (li < nl && lk[li].k <= 0) ++li;
    if (li < nl) {
        LinkView cur, nxt;
        open_link(li, cur);
        load_tile(cur, 0);
        int img = 0;                 // DB: the image the MFMAs read in the current k-tile
        if (DB) {
            store_tile(cur, As, Bs);
            __syncthreads();
        }
        while (true) {
            // the next non-empty link is opened BEFORE the k loop of this one, so that the last k-tile of this link can
            // prefetch the first k-tile of the next (no drained load pipeline at a link boundary) while everything the k loop
            // uses from `cur` (LDS strides, fragment addresses) stays loop-invariant
            int lj = li + 1;
            while (lj < nl && lk[lj].k <= 0) ++lj;
            const bool has_next = lj < nl;
            if (has_next) open_link(lj, nxt);
            const int K = cur.K;
            const int sa_i = cur.sa_i, sa_k = cur.sa_k, sb_j = cur.sb_j, sb_k = cur.sb_k;
            const int fa = (wr * TM * 16 + l15) * sa_i + l4 * sa_k, fb = (wc * TN * 16 + l15) * sb_j + l4 * sb_k;
            if (DB) {
                for (int k0 = 0; k0 < K; k0 += BK) {
                    const bool more = (k0 + BK < K);
                    if (more)
                        load_tile(cur, k0 + BK);      // global -> registers: the next k-tile of this link ...
                    else if (has_next)
                        load_tile(nxt, 0);            // ... or the first k-tile of the next one
                    mma_tile(As + img * IMG + fa, Bs + img * IMG + fb, sa_i, sa_k, sb_j, sb_k);
                    // registers -> the OTHER image (nobody reads it: every wavefront passed the barrier that ended the k-tile which did)
                    if (more)
                        store_tile(cur, As + (img ^ 1) * IMG, Bs + (img ^ 1) * IMG);
                    else if (has_next)
                        store_tile(nxt, As + (img ^ 1) * IMG, Bs + (img ^ 1) * IMG);
                    __syncthreads();
                    img ^= 1;
                }
            } else {
                for (int k0 = 0; k0 < K; k0 += BK) {
                    store_tile(cur, As, Bs);
                    __syncthreads();
                    if (k0 + BK < K)
                        load_tile(cur, k0 + BK);      // prefetch the next k-tile into registers
                    else if (has_next)
                        load_tile(nxt, 0);            // ... or the first k-tile of the next link
                    mma_tile(As + fa, Bs + fb, sa_i, sa_k, sb_j, sb_k);
                    __syncthreads();
                }
            }
            if (!has_next) break;
            cur = nxt;
            li = lj;
        }
    }

    // ---- epilogue: C/D layout of v_mfma_f64_16x16x4_f64: col = lane&15, row = (lane>>4) + 4*reg
    double *Cp = Cbase + ES * tk.c_off;
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

static int g_large_variant = 1;  // 0: 4 waves x (64x64), 1: 8 waves x (64x32); test/tuning hook

extern "C" int tpa_gemm_set_variant(int v) {
    g_large_variant = v;
    return 0;
}

// cfg 0: 128 x 128 tiles (large blocks), cfg 1: 64 x 64 tiles (many small blocks / not enough tiles to
// fill 256 CUs).  complex: 128 x 64 and 64 x 32.
extern "C" int tpa_gemm_tile_shape(int dtype, int cfg, int *bm, int *bn) {
    if (dtype == TPA_F64) {
        *bm = cfg ? 64 : 128;
        *bn = cfg ? 64 : 128;
    } else {
        *bm = cfg ? 64 : 128;
        *bn = cfg ? 32 : 64;
    }
    return 0;
}

template <bool CPLX, int BM, int BN, int TM, int TN, int BK_, bool DB = false>
static void launch(const int64_t *tasks_dev, const int64_t *links_dev, const int32_t *tiles_dev, int n_tiles,
                   const void *Abase, const void *Bbase, void *Cbase, hipStream_t st) {
    using C = Cfg<CPLX, BM, BN, TM, TN, BK_>;
    gemm_chain_kernel<CPLX, BM, BN, TM, TN, BK_, DB><<<n_tiles, C::NT, 0, st>>>(
        (const Task *)tasks_dev, (const Link *)links_dev, (const int4 *)tiles_dev, (const double *)Abase,
        (const double *)Bbase, (double *)Cbase);
}

extern "C" int tpa_gemm_chain(int dtype, int cfg, const int64_t *tasks_dev, const int64_t *links_dev,
                              const int32_t *tiles_dev, int n_tiles, const void *Abase,
                              const void *Bbase, void *Cbase, void *stream) {
    TPA_ARG_CHECK(dtype == TPA_F64 || dtype == TPA_C128);
    TPA_ARG_CHECK(cfg == 0 || cfg == 1);
    if (n_tiles <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == TPA_F64) {
        if (cfg == 1 && (g_large_variant & 4))       // one wavefront owns the whole 64 x 64 tile (4 x 4 MFMA tiles: 0.5 LDS reads per MFMA)
            launch<false, 64, 64, 4, 4, 16>(tasks_dev, links_dev, tiles_dev, n_tiles, Abase, Bbase, Cbase, st);
        else if (cfg == 1 && (g_large_variant & 8))  // two wavefronts, 64 x 32 each (4 x 2: 0.75 reads per MFMA)
            launch<false, 64, 64, 4, 2, 16>(tasks_dev, links_dev, tiles_dev, n_tiles, Abase, Bbase, Cbase, st);
        else if (cfg == 1 && (g_large_variant & 16))  // eight wavefronts, 16 x 32 each (1 x 2: 1.5 reads per MFMA, twice the waves per tile)
            launch<false, 64, 64, 1, 2, 16>(tasks_dev, links_dev, tiles_dev, n_tiles, Abase, Bbase, Cbase, st);
        else if (cfg == 1 && (g_large_variant & 32))  // sixteen wavefronts, one MFMA tile each
            launch<false, 64, 64, 1, 1, 16>(tasks_dev, links_dev, tiles_dev, n_tiles, Abase, Bbase, Cbase, st);
        else if (cfg == 1 && (g_large_variant & 2))
            launch<false, 64, 64, 2, 2, 32>(tasks_dev, links_dev, tiles_dev, n_tiles, Abase, Bbase, Cbase, st);
        else if (cfg == 1 && n_tiles <= 1024 && !(g_large_variant & 64))
            // few tiles (chi <= 512, ramp sweeps, edge bonds): every tile is resident at once and the launch lasts as long as ONE tile's
            // chain -- eight wavefronts per tile (16 x 32 each) halve that chain.  Measured (scripts/gemm_bench.py): matvec at chi = 512
            // 6.9 -> 8.0 TFLOP/s (step 2, 69 tiles: 0.089 -> 0.075 ms), at chi = 2048 step 2 (831 tiles) 0.710 -> 0.687 ms, step 1 (4005 tiles) 0.660 -> 0.680 ms: hence the limit of 1024 tiles.  The opposite
            // direction -- ONE wavefront per 64 x 64 tile (4 x 4 MFMA tiles, half the LDS reads per MFMA) or two (4 x 2) -- is slower
            // everywhere: 24.6 / 30.8 instead of 38.7 TFLOP/s at chi = 2048: the tile's chain, not the LDS bandwidth, is the limit.
            (g_large_variant & 128) ? launch<false, 64, 64, 1, 2, 16, true>(tasks_dev, links_dev, tiles_dev, n_tiles, Abase, Bbase, Cbase, st)
                                    : launch<false, 64, 64, 1, 2, 16>(tasks_dev, links_dev, tiles_dev, n_tiles, Abase, Bbase, Cbase, st);
        else if (cfg == 1)
            // the default of the DMRG path.  Bit 7 (128) of tpa_gemm_set_variant switches the double-buffered loop ON (DB, round 5).
            // Measured on the MI355X (scripts/gemm_bench.py, profiles/r05_gemm_double_buffer.txt): dense 4096^3 on 64 x 64 tiles 50.0 ->
            // 46.2 TFLOP/s, matvec at chi = 2048 38.6 -> 35.5, at chi = 512 8.0 -> 6.8: SLOWER.  The second image takes the workgroups
            // per CU from six (registers) to four (40 KB of LDS each), and the loop is bound by latency hiding across workgroups, not
            // by its two barriers -- the same finding as the BK = 32 variant of round 3 (half the barriers, lower occupancy, slower).
            (g_large_variant & 128) ? launch<false, 64, 64, 2, 2, 16, true>(tasks_dev, links_dev, tiles_dev, n_tiles, Abase, Bbase, Cbase, st)
                                    : launch<false, 64, 64, 2, 2, 16>(tasks_dev, links_dev, tiles_dev, n_tiles, Abase, Bbase, Cbase, st);
        else if (g_large_variant & 1)
            launch<false, 128, 128, 4, 2, 16>(tasks_dev, links_dev, tiles_dev, n_tiles, Abase, Bbase, Cbase, st);
        else
            launch<false, 128, 128, 4, 4, 16>(tasks_dev, links_dev, tiles_dev, n_tiles, Abase, Bbase, Cbase, st);
    } else {
        if (cfg == 1)
            launch<true, 64, 32, 2, 1, 16>(tasks_dev, links_dev, tiles_dev, n_tiles, Abase, Bbase, Cbase, st);
        else
            launch<true, 128, 64, 4, 2, 16>(tasks_dev, links_dev, tiles_dev, n_tiles, Abase, Bbase, Cbase, st);
    }
    TPA_LAUNCH_CHECK();
    return 0;
}
