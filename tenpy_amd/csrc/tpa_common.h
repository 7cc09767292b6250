// Shared helpers for the tenpy_amd HIP sources (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/tenpy_amd.h"

#define TPA_WAVE 64

extern thread_local char tpa_errbuf[512];

#define TPA_HIP_CHECK(expr)                                                                 \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            snprintf(tpa_errbuf, sizeof(tpa_errbuf), "%s:%d: %s -> %s", __FILE__, __LINE__, \
                     #expr, hipGetErrorString(_e));                                         \
            return (int)_e;                                                                 \
        }                                                                                   \
    } while (0)

#define TPA_LAUNCH_CHECK() TPA_HIP_CHECK(hipGetLastError())

#define TPA_ARG_CHECK(cond)                                                                   \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            snprintf(tpa_errbuf, sizeof(tpa_errbuf), "%s:%d: bad argument: %s", __FILE__,     \
                     __LINE__, #cond);                                                        \
            return TPA_E_BADARG;                                                              \
        }                                                                                     \
    } while (0)

struct cplx {
    double re, im;
};

// Cross-lane moves inside a row of 16 lanes on the DPP path of the VALU (no LDS round trip, ~4x cheaper than ds_bpermute):
// quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140.
// All 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ int dpp_mov_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    const int lo = dpp_mov_i32<CTRL>(__double2loint(v)), hi = dpp_mov_i32<CTRL>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Sum / max / min over the 64 lanes; the result is bit-identical in every lane (each step combines a lane with its mirror
// image, and the operations are commutative).  Four DPP steps inside the rows, two ds_bpermute steps across them.
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_mov_f64<0xB1>(v);
    v += dpp_mov_f64<0x4E>(v);
    v += dpp_mov_f64<0x141>(v);
    v += dpp_mov_f64<0x140>(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_mov_f64<0xB1>(v));
    v = fmax(v, dpp_mov_f64<0x4E>(v));
    v = fmax(v, dpp_mov_f64<0x141>(v));
    v = fmax(v, dpp_mov_f64<0x140>(v));
    v = fmax(v, __shfl_xor(v, 16, 64));
    v = fmax(v, __shfl_xor(v, 32, 64));
    return v;
}
__device__ __forceinline__ int wave_min(int v) {
    v = min(v, dpp_mov_i32<0xB1>(v));
    v = min(v, dpp_mov_i32<0x4E>(v));
    v = min(v, dpp_mov_i32<0x141>(v));
    v = min(v, dpp_mov_i32<0x140>(v));
    v = min(v, __shfl_xor(v, 16, 64));
    v = min(v, __shfl_xor(v, 32, 64));
    return v;
}

// Workgroup barrier that orders LDS traffic only.  `__syncthreads()` also waits for every outstanding GLOBAL store of the
// wavefront (vmcnt(0) of its workgroup-scope fence): in a latency chain that stores results to memory as it goes (panel
// factorisation) each barrier would then cost a full store round trip.  Use only where no thread reads global data that another
// thread of the workgroup wrote earlier in the same kernel.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Sum over a workgroup of NT threads; result valid in every thread. `red` >= NT/64 doubles of LDS.
template <int NT>
__device__ __forceinline__ double block_sum(double v, double *red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    double t = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}
