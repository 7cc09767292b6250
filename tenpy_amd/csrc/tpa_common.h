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

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
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
