// Latency of a dependent launch (a) on the same stream, (b) on another stream through an event (hipStreamWaitEvent), and
// (c) two independent ~25 us kernels per step on two streams with cross dependencies every step (the shape of a look-ahead QR:
// panel(s) || update(s), both waiting for panel(s-1) and update(s-1)).    hipcc --offload-arch=gfx950 -O3 stream_hop.hip -o stream_hop
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void spin_kernel(double *p, int iters) {
    double v = p[threadIdx.x & 63];
    for (int i = 0; i < iters; ++i) v = fma(v, 1.0000001, 1e-9);
    if (v == 12345.678) p[0] = v;
}

int main() {
    double *d;
    CK(hipMalloc(&d, 4096));
    CK(hipMemset(d, 0, 4096));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    const int N = 400;
    std::vector<hipEvent_t> ea(N), eb(N);
    for (int i = 0; i < N; ++i) {
        CK(hipEventCreateWithFlags(&ea[i], hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&eb[i], hipEventDisableTiming));
    }
    for (int iters : {0, 6000}) {      // 0: empty kernel (pure launch chain); 6000: ~25 us of dependent FMAs
        for (int rep = 0; rep < 2; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) spin_kernel<<<1, 64, 0, a>>>(d, iters);
            CK(hipStreamSynchronize(a));
            double same = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {          // ping-pong: every launch waits for the previous one on the OTHER stream
                hipStream_t s = (i & 1) ? b : a;
                if (i > 0) CK(hipStreamWaitEvent(s, (i & 1) ? ea[i - 1] : eb[i - 1], 0));
                spin_kernel<<<1, 64, 0, s>>>(d, iters);
                CK(hipEventRecord((i & 1) ? eb[i] : ea[i], s));
            }
            CK(hipStreamSynchronize(a));
            CK(hipStreamSynchronize(b));
            double hop = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {          // two kernels per step, one per stream, each waits for BOTH kernels of the previous step
                if (i > 0) {
                    CK(hipStreamWaitEvent(a, eb[i - 1], 0));
                    CK(hipStreamWaitEvent(b, ea[i - 1], 0));
                }
                spin_kernel<<<1, 64, 0, a>>>(d, iters);
                spin_kernel<<<1, 64, 0, b>>>(d, iters);
                CK(hipEventRecord(ea[i], a));
                CK(hipEventRecord(eb[i], b));
            }
            CK(hipStreamSynchronize(a));
            CK(hipStreamSynchronize(b));
            double pair = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            if (rep == 1)
                printf("iters %5d: same stream %.2f us per launch | ping-pong over two streams %.2f us per launch | lock-step pair %.2f us per step (two kernels)\n",
                       iters, same, hop, pair);
        }
    }
    return 0;
}
