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

// IDENTITY-ROW SKIP (round 6; used by the block SVD's product [W | G] <- Qtot [W | G], tpa_svd.hip: svd_run): a tile with tile.w > 0 of a
// task with pad != 0 looks at the word ((const int *)pad)[tile.w - 1]; 0 means "the rows of A that this tile of C needs are unit vectors
// e_row" (those rows of Qtot did not rotate in the sweep), so C[row, :] = B[row, :] and the tile is a COPY of the first link's B operand
// instead of a chain of K / 16 k-tiles.  Every other caller leaves tile.w = 0 and pad = 0.  Returns true if the tile was handled.
template <int ES, int NT, int BM, int BN>
__device__ __forceinline__ bool gemm_identity_rows_tile(const Task &tk, const Link *__restrict__ links, const int4 tile, const double *__restrict__ Bbase,
                                                        double *__restrict__ Cbase, const int tid) {
    if (tile.w <= 0 || tk.pad == 0) return false;
    if (reinterpret_cast<const int *>(tk.pad)[tile.w - 1] != 0) return false;
    const Link L = links[tk.link_begin];
    const int row0 = tile.y * BM, col0 = tile.z * BN;
    const int rows = min(BM, (int)tk.m - row0), cols = min(BN, (int)tk.n - col0);
    for (int e = tid; e < rows * BN; e += NT) {
        const int r = e / BN, c = e % BN;
        if (c < cols) {
            const int64_t src = L.b_off + (int64_t)(row0 + r) * L.b_ks + (int64_t)(col0 + c) * L.b_ns;
            const int64_t dst = tk.c_off + (int64_t)(row0 + r) * tk.ldc + col0 + c;
#pragma unroll
            for (int q = 0; q < ES; ++q) Cbase[ES * dst + q] = Bbase[ES * src + q];
        }
    }
    return true;
}

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

template <bool CPLX, int BM, int BN, int TM, int TN, int BK_>
__global__ __launch_bounds__((Cfg<CPLX, BM, BN, TM, TN, BK_>::NT)) void gemm_chain_kernel(
    const Task *__restrict__ tasks, const Link *__restrict__ links, const int4 *__restrict__ tiles,
    const double *__restrict__ Abase, const double *__restrict__ Bbase, double *__restrict__ Cbase) {
    using C = Cfg<CPLX, BM, BN, TM, TN, BK_>;
    constexpr int NT = C::NT, EA = C::EA, EB = C::EB, PL = C::PLANES, BK = C::BK;
    constexpr int ES = CPLX ? 2 : 1;  // doubles per element
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
    if (gemm_identity_rows_tile<ES, NT, BM, BN>(tk, links, tile, Bbase, Cbase, tid)) return;      // (workgroup-uniform)

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


// ---------------------------------------------------------------------------------------------------------------------------------
// Round 6: the real fp64 kernel of the large launches.  Same tables, same chain semantics, different loop:
//  * a wavefront owns TM x TN = 4 x 4 (or 4 x 2) MFMA tiles: 8 (6) fragment reads per 16 (8) MFMAs instead of 4 per 4;
//  * fragments are double-buffered in registers: the ds_reads of k-step s + 1 are in flight while the MFMAs of step s issue
//    (the old loop re-used the fragment registers and so exposed one LDS round trip per k-step);
//  * two LDS images: the registers -> LDS stores of k-tile t + 1 and the global loads of k-tile t + 2 are issued BETWEEN the MFMA
//    groups of k-tile t, and ONE barrier per k-tile is left;
//  * one LDS layout ([row][BK + 1], conflict-free for both the k-fast and the row-fast global order: 34 i + 2 kk hits 32 distinct
//    bank pairs per half-wave either way), so every fragment read is base + immediate;
//  * full k-tiles are loaded WITHOUT predicates: a row / column index beyond the block is clamped to its last row (the products it
//    feeds belong to rows / columns of C that are never stored); only the last partial k-tile of a link tests k < K per element;
//  * a wavefront whose sub-tile lies wholly outside the block skips its MFMAs (64-row granularity of the arithmetic on 128-row tiles).
template <int BM, int BN, int TM, int TN>
struct Cfg2 {
    static constexpr int BK = 16, LD = BK + 1;
    static constexpr int WM = BM / (TM * 16), WN = BN / (TN * 16), NT = WM * WN * 64;
    static constexpr int A_LDS = BM * LD, B_LDS = BN * LD, IMG = A_LDS + B_LDS;       // doubles
    static constexpr int EA = BM * BK / NT, EB = BN * BK / NT;
    static constexpr int LDS_BYTES = 2 * IMG * 8;
    static_assert(NT % BK == 0 && NT % BM == 0 && NT % BN == 0, "staging offsets must be affine in the element index");
};

template <int BM, int BN, int TM, int TN>
__global__ __launch_bounds__((Cfg2<BM, BN, TM, TN>::NT)) __attribute__((amdgpu_waves_per_eu(TM * TN >= 16 ? 2 : (TM * TN >= 8 ? 3 : 4))))
void gemm_chain2_kernel(
    const Task *__restrict__ tasks, const Link *__restrict__ links, const int4 *__restrict__ tiles,
    const double *__restrict__ Abase, const double *__restrict__ Bbase, double *__restrict__ Cbase) {
    using C = Cfg2<BM, BN, TM, TN>;
    constexpr int NT = C::NT, EA = C::EA, EB = C::EB, BK = C::BK, LD = C::LD, IMG = C::IMG;
    extern __shared__ double lds2[];

    const int4 tile = tiles[blockIdx.x];
    const Task tk = tasks[tile.x];
    const int m = (int)tk.m, n = (int)tk.n;
    const int row0 = tile.y * BM, col0 = tile.z * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (scalar: the skip tests below are branches, not masks)
    const int wr = wave / C::WN, wc = wave % C::WN;
    const int l15 = lane & 15, l4 = lane >> 4;
    // a wavefront whose sub-tile lies wholly outside the block stages its share of the operands and skips the arithmetic
    const bool wave_act = (row0 + wr * TM * 16 < m) && (col0 + wc * TN * 16 < n);
    if (gemm_identity_rows_tile<1, NT, BM, BN>(tk, links, tile, Bbase, Cbase, tid)) return;      // (workgroup-uniform)

    d4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = d4{0, 0, 0, 0};

    const int nl = (int)tk.link_count;
    const Link *lk = links + tk.link_begin;

    // what a thread needs to stage its share of the k-tiles of one link
    struct View {
        const double *pa, *pb;       // advance by one k-tile per load
        int64_t ka, kb;              // that advance (doubles)
        int64_t oa[EA], ob[EB];      // element offsets of the thread's elements relative to pa / pb (rows / columns clamped into the block)
        int kka, dka, kkb, dkb;      // k index inside the tile of element r: kk + r * dk (tail predicate)
        int la, dla, lb, dlb;        // LDS offset of element r: l + r * dl (doubles, inside an operand plane)
        int K;
    };
    auto open_link = [&](int li, View &v) {
        const Link L = lk[li];
        v.K = (int)L.k;
        const bool a_kfast = (L.a_ks == 1);
        const bool b_kfast = (L.b_ks == 1) && (L.b_ns != 1);
        {   // element e = tid + r NT;  k-fast: (i, kk) = (e / BK, e % BK);  else (e % BM, e / BM)
            const int i = a_kfast ? (tid / BK) : (tid % BM), kk = a_kfast ? (tid % BK) : (tid / BM);
            const int di = a_kfast ? (NT / BK) : 0, dk = a_kfast ? 0 : (NT / BM);
            v.pa = Abase + L.a_off;
            v.ka = (int64_t)BK * L.a_ks;
            v.kka = kk;
            v.dka = dk;
            v.la = i * LD + kk;
            v.dla = di * LD + dk;
#pragma unroll
            for (int r = 0; r < EA; ++r)
                v.oa[r] = (int64_t)min(row0 + i + r * di, m - 1) * L.a_rs + (int64_t)(kk + r * dk) * L.a_ks;
        }
        {
            const int j = b_kfast ? (tid / BK) : (tid % BN), kk = b_kfast ? (tid % BK) : (tid / BN);
            const int dj = b_kfast ? (NT / BK) : 0, dk = b_kfast ? 0 : (NT / BN);
            v.pb = Bbase + L.b_off;
            v.kb = (int64_t)BK * L.b_ks;
            v.kkb = kk;
            v.dkb = dk;
            v.lb = j * LD + kk;
            v.dlb = dj * LD + dk;
#pragma unroll
            for (int r = 0; r < EB; ++r)
                v.ob[r] = (int64_t)min(col0 + j + r * dj, n - 1) * L.b_ns + (int64_t)(kk + r * dk) * L.b_ks;
        }
    };
    double ra[EA], rb[EB];
    // registers <- the next k-tile of link v (krem = K - k0 >= 1 elements of k are left); advances the link's pointers
    auto load_tile = [&](View &v, int krem) {
        if (krem >= BK) {
#pragma unroll
            for (int r = 0; r < EA; ++r) ra[r] = v.pa[v.oa[r]];
#pragma unroll
            for (int r = 0; r < EB; ++r) rb[r] = v.pb[v.ob[r]];
        } else {
#pragma unroll
            for (int r = 0; r < EA; ++r) ra[r] = (v.kka + r * v.dka < krem) ? v.pa[v.oa[r]] : 0.0;
#pragma unroll
            for (int r = 0; r < EB; ++r) rb[r] = (v.kkb + r * v.dkb < krem) ? v.pb[v.ob[r]] : 0.0;
        }
        v.pa += v.ka;
        v.pb += v.kb;
    };
    auto store_tile = [&](const View &v, double *img) {
#pragma unroll
        for (int r = 0; r < EA; ++r) img[v.la + r * v.dla] = ra[r];
#pragma unroll
        for (int r = 0; r < EB; ++r) img[C::A_LDS + v.lb + r * v.dlb] = rb[r];
    };
    const int fa0 = (wr * TM * 16 + l15) * LD + l4, fb0 = C::A_LDS + (wc * TN * 16 + l15) * LD + l4;
    double fa[2][TM], fb[2][TN];
    auto read_frag = [&](const double *img, int ks, int buf) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[buf][i] = img[fa0 + i * 16 * LD + ks * 4];
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[buf][j] = img[fb0 + j * 16 * LD + ks * 4];
    };
    auto mma_step = [&](int buf) {
        if (wave_act) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[buf][i], fb[buf][j], acc[i][j], 0, 0, 0);
        }
    };

    // ---- the chain as one sequence of k-tiles: `st` = the link whose tile was loaded last (its pointers sit on the NEXT tile),
    //      lnext / knext = where the next load comes from
    View st;
    int li = 0;
    while (li < nl && lk[li].k <= 0) ++li;
    if (li < nl) {
        open_link(li, st);
        int kdone = 0;                     // k elements of link li already loaded
        auto load_next = [&]() -> bool {   // loads the next k-tile of the chain into the registers; false at the end of the chain
            if (kdone >= st.K) {
                int lj = li + 1;
                while (lj < nl && lk[lj].k <= 0) ++lj;
                if (lj >= nl) return false;
                li = lj;
                open_link(li, st);
                kdone = 0;
            }
            load_tile(st, st.K - kdone);
            kdone += BK;
            return true;
        };
        load_next();
        store_tile(st, lds2);
        bool have = load_next();           // registers: k-tile 1 (in flight)
        lds_barrier();
        int img = 0;
        while (true) {
            const double *cur = lds2 + img * IMG;
            double *oth = lds2 + (img ^ 1) * IMG;
            read_frag(cur, 0, 0);
            read_frag(cur, 1, 1);
            mma_step(0);
            // between the MFMA groups: k-tile t + 1 registers -> the other image (every wavefront left it at the last barrier),
            // then the global loads of k-tile t + 2
            const bool had = have;
            if (had) {
                store_tile(st, oth);
                have = load_next();
            }
            read_frag(cur, 2, 0);
            mma_step(1);
            read_frag(cur, 3, 1);
            mma_step(0);
            mma_step(1);
            if (!had) break;
            lds_barrier();
            img ^= 1;
        }
    }

    // ---- epilogue: C/D layout of v_mfma_f64_16x16x4_f64: col = lane&15, row = (lane>>4) + 4*reg
    double *Cp = Cbase + tk.c_off;
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
                    double v = acc[i][j][r];
                    if (accum) v += Cp[g];
                    Cp[g] = v;
                }
            }
}

}  // namespace

#include <stdlib.h>
static int g_variant = getenv("TPA_GEMM_VARIANT") ? atoi(getenv("TPA_GEMM_VARIANT")) : 0;  // tuning hook (see tpa_gemm_chain)

extern "C" int tpa_gemm_set_variant(int v) {
    g_variant = v;
    return 0;
}

// cfg 0: 128 x 64 tiles (large blocks), cfg 1: 64 x 64 tiles (many small blocks / not enough tiles to
// fill 256 CUs).  complex: 128 x 64 and 64 x 32.
extern "C" int tpa_gemm_tile_shape(int dtype, int cfg, int *bm, int *bn) {
    *bm = cfg ? 64 : 128;
    *bn = (dtype == TPA_F64) ? 64 : (cfg ? 32 : 64);
    return 0;
}

template <bool CPLX, int BM, int BN, int TM, int TN, int BK_>
static void launch(const int64_t *tasks_dev, const int64_t *links_dev, const int32_t *tiles_dev, int n_tiles,
                   const void *Abase, const void *Bbase, void *Cbase, hipStream_t st) {
    using C = Cfg<CPLX, BM, BN, TM, TN, BK_>;
    gemm_chain_kernel<CPLX, BM, BN, TM, TN, BK_><<<n_tiles, C::NT, 0, st>>>(
        (const Task *)tasks_dev, (const Link *)links_dev, (const int4 *)tiles_dev, (const double *)Abase,
        (const double *)Bbase, (double *)Cbase);
}

template <int BM, int BN, int TM, int TN>
static int launch2(const int64_t *tasks_dev, const int64_t *links_dev, const int32_t *tiles_dev, int n_tiles,
                   const void *Abase, const void *Bbase, void *Cbase, hipStream_t st) {
    using C = Cfg2<BM, BN, TM, TN>;
    static bool attr_set = false;      // more than 64 KB of LDS per workgroup must be asked for (once per instantiation)
    if (!attr_set) {
        TPA_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_chain2_kernel<BM, BN, TM, TN>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          C::LDS_BYTES));
        attr_set = true;
    }
    gemm_chain2_kernel<BM, BN, TM, TN><<<n_tiles, C::NT, C::LDS_BYTES, st>>>(
        (const Task *)tasks_dev, (const Link *)links_dev, (const int4 *)tiles_dev, (const double *)Abase,
        (const double *)Bbase, (double *)Cbase);
    return 0;
}

extern "C" int tpa_gemm_chain(int dtype, int cfg, const int64_t *tasks_dev, const int64_t *links_dev,
                              const int32_t *tiles_dev, int n_tiles, const void *Abase,
                              const void *Bbase, void *Cbase, void *stream) {
    TPA_ARG_CHECK(dtype == TPA_F64 || dtype == TPA_C128);
    TPA_ARG_CHECK(cfg == 0 || cfg == 1);
    if (n_tiles <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    int rc = 0;
    if (dtype == TPA_F64) {
        // Measured on the MI355X (scripts/gemm_bench.py with REPS=40 -- the clock needs ~10 ms of load to settle, shorter timings read
        // 8 % low; profiles/r06_gemm_v2.txt).  Dense 4096^3: 128 x 64 tiles 59.0 TFLOP/s (the old loop: 49.8 on 64 x 64, 47.1 on 128 x 128 tiles),
        // 128 x 128 tiles with 4 x 4 MFMA tiles per wavefront 52.7 (two workgroups per CU by registers; the 128 x 64 tile has three).
        // Lanczos matvec on the chi = 2048 Sz structure: 64 x 64 tiles 48.3 TFLOP/s (old loop 45.6), 128 x 64 46.2, 128 x 128 43.9 -- the
        // block edges (871, 450, 148 ... rows) pad less on the small tile.  Few tiles (<= 1024, all resident at once: chi <= 512, edge
        // bonds): the launch lasts as long as ONE tile's chain, eight wavefronts per tile (16 x 32 each, the round-4 loop) halve it; the new loop
        // on 2 x 2 or 1 x 2 MFMA tiles per wavefront is 5 % slower there (chi = 512: 12.8 / 13.5 against 13.5 TFLOP/s).  g_variant bit 0 = new loop everywhere.
        if (cfg == 0)
            rc = launch2<128, 64, 4, 2>(tasks_dev, links_dev, tiles_dev, n_tiles, Abase, Bbase, Cbase, st);
        else if (n_tiles <= 1024 && !(g_variant & 1))
            launch<false, 64, 64, 1, 2, 16>(tasks_dev, links_dev, tiles_dev, n_tiles, Abase, Bbase, Cbase, st);
        else
            rc = launch2<64, 64, 2, 2>(tasks_dev, links_dev, tiles_dev, n_tiles, Abase, Bbase, Cbase, st);
    } else {
        if (cfg == 1)
            launch<true, 64, 32, 2, 1, 16>(tasks_dev, links_dev, tiles_dev, n_tiles, Abase, Bbase, Cbase, st);
        else
            launch<true, 128, 64, 4, 2, 16>(tasks_dev, links_dev, tiles_dev, n_tiles, Abase, Bbase, Cbase, st);
    }
    if (rc) return rc;
    TPA_LAUNCH_CHECK();
    return 0;
}
