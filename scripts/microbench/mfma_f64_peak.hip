// Measures the achievable v_mfma_f64_16x16x4_f64 rate on the device (the roofline "peak" we can actually
// reach, incl. clock behaviour): NW waves per CU workgroup, NACC independent accumulators, no memory traffic.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(double *out, int iters, double a0, double b0) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int nthreads, int blocks_per_cu) {
    int ncu = 256;
    int grid = ncu * blocks_per_cu;
    double *out;
    hipMalloc(&out, sizeof(double) * grid * nthreads);
    int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<NACC><<<grid, nthreads>>>(out, 100, 1.0, 0.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<grid, nthreads>>>(out, iters, 1.0, 0.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * (nthreads / 64) * iters * NACC * 2048.0;
    printf("threads/WG=%d WG/CU=%d NACC=%d : %.2f ms  %.1f TFLOP/s\n", nthreads, blocks_per_cu, NACC, ms, flops / ms / 1e9);
    hipFree(out);
}
int main() {
    run<4>(256, 1);
    run<16>(256, 1);
    run<16>(256, 2);
    run<16>(512, 1);
    run<8>(1024, 1);
    return 0;
}
