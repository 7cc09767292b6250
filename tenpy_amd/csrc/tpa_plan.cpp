// Host planner: the integer bookkeeping of a block-sparse tensordot.
//
// Restates what the reference's _tensordot_worker does before it fills its GEMM batch
// (tenpy/linalg/_npc_helper.pyx:1498-1786: _tensordot_pre_sort :1337, _find_row_differences_qdata :671,
// _tensordot_match_charges :1382, _iter_common_sorted_push :1299) with a different algorithm: a hash
// join on the fused contracted qindex instead of sorted row/column lists + charge matching.  Two
// blocks that share a contracted qindex tuple automatically have compatible charges (each operand
// obeys its own charge rule), so no charge arithmetic is needed here; the result block list and its
// lexsorted order (last leg most significant, :1777 `res._qdata_sorted = True`) are identical.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <unordered_map>
#include <vector>

#include "../../include/tenpy_amd.h"

thread_local char tpa_errbuf[512] = {0};

extern "C" const char *tpa_last_error(void) { return tpa_errbuf; }
extern "C" int tpa_version(void) { return 100; }

namespace {

// rank each block's "keep" tuple in lexsorted-unique order (last entry most significant)
void rank_keep(const int64_t *qdata, int64_t nblk, int rank, int keep_begin, int keep_len,
               std::vector<int64_t> &id_of_block, std::vector<int64_t> &first_block) {
    std::vector<int64_t> order(nblk);
    for (int64_t i = 0; i < nblk; ++i) order[i] = i;
    auto cmp = [&](int64_t x, int64_t y) {
        for (int d = keep_len - 1; d >= 0; --d) {
            const int64_t vx = qdata[x * rank + keep_begin + d], vy = qdata[y * rank + keep_begin + d];
            if (vx != vy) return vx < vy;
        }
        return false;
    };
    std::stable_sort(order.begin(), order.end(), cmp);
    id_of_block.assign(nblk, 0);
    first_block.clear();
    int64_t id = -1;
    for (int64_t s = 0; s < nblk; ++s) {
        if (s == 0 || cmp(order[s - 1], order[s])) {
            ++id;
            first_block.push_back(order[s]);
        }
        id_of_block[order[s]] = id;
    }
}

struct Emit {
    int64_t col, row, ckey, a, b;
};

}  // namespace

extern "C" int tpa_plan_tensordot(const int64_t *a_qdata, int64_t na, int ra, const int64_t *b_qdata,
                                  int64_t nb, int rb, int ncontr, const int64_t *contr_nblocks,
                                  int64_t *res_qdata, int64_t *res_a_first, int64_t *res_b_first,
                                  int64_t cap_res, int64_t *n_res, int64_t *gemm, int64_t cap_gemm,
                                  int64_t *n_gemm) {
    if (ncontr < 0 || ncontr > ra || ncontr > rb || !n_res || !n_gemm) {
        snprintf(tpa_errbuf, sizeof(tpa_errbuf), "tpa_plan_tensordot: bad ranks");
        return TPA_E_BADARG;
    }
    const int ka = ra - ncontr, kb = rb - ncontr;
    std::vector<int64_t> row_of, row_first, col_of, col_first;
    rank_keep(a_qdata, na, ra, 0, ka, row_of, row_first);
    rank_keep(b_qdata, nb, rb, ncontr, kb, col_of, col_first);
    // fused contracted key (mixed radix, first contracted leg fastest; any bijection works)
    auto ckey_a = [&](int64_t i) {
        int64_t key = 0, mul = 1;
        for (int d = 0; d < ncontr; ++d) {
            key += a_qdata[i * ra + ka + d] * mul;
            mul *= contr_nblocks[d];
        }
        return key;
    };
    auto ckey_b = [&](int64_t i) {
        int64_t key = 0, mul = 1;
        for (int d = 0; d < ncontr; ++d) {
            key += b_qdata[i * rb + d] * mul;
            mul *= contr_nblocks[d];
        }
        return key;
    };
    std::unordered_map<int64_t, std::vector<int64_t>> by_key;
    by_key.reserve((size_t)nb * 2 + 1);
    for (int64_t j = 0; j < nb; ++j) by_key[ckey_b(j)].push_back(j);
    std::vector<Emit> em;
    em.reserve((size_t)na * 2);
    for (int64_t i = 0; i < na; ++i) {
        const int64_t key = ckey_a(i);
        auto it = by_key.find(key);
        if (it == by_key.end()) continue;
        for (int64_t j : it->second) em.push_back(Emit{col_of[j], row_of[i], key, i, j});
    }
    std::sort(em.begin(), em.end(), [](const Emit &x, const Emit &y) {
        if (x.col != y.col) return x.col < y.col;
        if (x.row != y.row) return x.row < y.row;
        return x.ckey < y.ckey;
    });
    int64_t nres = 0;
    const int64_t ngemm = (int64_t)em.size();
    for (int64_t g = 0; g < ngemm; ++g)
        if (g == 0 || em[g].col != em[g - 1].col || em[g].row != em[g - 1].row) ++nres;
    *n_res = nres;
    *n_gemm = ngemm;
    if (nres > cap_res || ngemm > cap_gemm) {
        snprintf(tpa_errbuf, sizeof(tpa_errbuf), "tpa_plan_tensordot: capacity too small (need %lld res, %lld gemm)",
                 (long long)nres, (long long)ngemm);
        return TPA_E_BADARG;
    }
    const int rr = ka + kb;
    int64_t r = -1;
    for (int64_t g = 0; g < ngemm; ++g) {
        if (g == 0 || em[g].col != em[g - 1].col || em[g].row != em[g - 1].row) {
            ++r;
            for (int d = 0; d < ka; ++d) res_qdata[r * rr + d] = a_qdata[em[g].a * ra + d];
            for (int d = 0; d < kb; ++d) res_qdata[r * rr + ka + d] = b_qdata[em[g].b * rb + ncontr + d];
            res_a_first[r] = em[g].a;
            res_b_first[r] = em[g].b;
        }
        gemm[3 * g + 0] = r;
        gemm[3 * g + 1] = em[g].a;
        gemm[3 * g + 2] = em[g].b;
    }
    return 0;
}
