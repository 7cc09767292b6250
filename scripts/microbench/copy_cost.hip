// What does a small hipMemsetAsync / hipMemcpyAsync cost in a stream of dependent small kernels (the launch chains of the block SVD)?
// Per element of a chain of N: [empty kernel, X] with X = nothing | 64-B memset | own zero-fill kernel | 256-B H2D from pageable memory |
// the same from pinned memory | 16-B D2H into pinned memory | a 1-thread kernel writing 16 B to mapped pinned memory.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void empty_kernel(double *p) {
    if (p[0] == 12345.678) p[1] = 1.;
}
__global__ void zero_kernel(double *p, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0.;
}
__global__ void post_kernel(const double *src, double *host) {
    host[0] = src[0];
    host[1] = src[1];
    __threadfence_system();
}

int main() {
    double *d, *pinned;
    CK(hipMalloc(&d, 1 << 16));
    CK(hipMemset(d, 0, 1 << 16));
    CK(hipHostMalloc((void **)&pinned, 4096, hipHostMallocDefault));
    std::vector<double> pageable(512, 1.);
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int N = 500;
    const char *names[] = {"kernel only", "+ hipMemsetAsync 64 B", "+ own zero kernel 64 B", "+ H2D 256 B pageable", "+ H2D 256 B pinned",
                           "+ D2H 16 B to pinned", "+ post kernel to mapped pinned"};
    for (int mode = 0; mode < 7; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipStreamSynchronize(s));
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {
                empty_kernel<<<1, 64, 0, s>>>(d);
                switch (mode) {
                case 1: CK(hipMemsetAsync(d + 64, 0, 64, s)); break;
                case 2: zero_kernel<<<1, 64, 0, s>>>(d + 64, 8); break;
                case 3: CK(hipMemcpyAsync(d + 128, pageable.data(), 256, hipMemcpyHostToDevice, s)); break;
                case 4: CK(hipMemcpyAsync(d + 128, pinned + 64, 256, hipMemcpyHostToDevice, s)); break;
                case 5: CK(hipMemcpyAsync(pinned, d, 16, hipMemcpyDeviceToHost, s)); break;
                case 6: post_kernel<<<1, 1, 0, s>>>(d, pinned); break;
                default: break;
                }
            }
            auto t1 = std::chrono::steady_clock::now();
            CK(hipStreamSynchronize(s));
            auto t2 = std::chrono::steady_clock::now();
            if (rep == 1)
                printf("%-34s host enqueue %.2f us per element, total %.2f us per element\n", names[mode],
                       std::chrono::duration<double, std::micro>(t1 - t0).count() / N, std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
        }
    return 0;
}
