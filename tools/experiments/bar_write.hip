// bar_write.hip -- can the host write a gradient straight into device memory (fine-grained allocation mapped through the PCIe
// BAR), and what does it cost against "memcpy into a pinned buffer + a kernel that pulls it" (k_stage) for 128 KiB?
// hipcc --offload-arch=gfx950 -O2 -o bar_write bar_write.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { std::printf("%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)
__global__ void k_sum(const double* p, long long n, double* out) {
    double s = 0.0;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) s += p[i];
    __shared__ double sh[256];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0; for (int i = 0; i < 256; ++i) t += sh[i]; *out = t; }
}
__global__ void k_pull(const double* h, double* d, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) d[i] = h[i];
}
int main() {
    const long long n = 16384;
    std::vector<double> src(n);
    for (long long i = 0; i < n; ++i) src[i] = 1.0 + 1e-6 * i;
    double *dfine = nullptr, *dout = nullptr, *dplain = nullptr, *hpin = nullptr;
    CK(hipMalloc(&dout, 8));
    CK(hipMalloc(&dplain, n * 8));
    CK(hipHostMalloc(&hpin, n * 8, hipHostMallocCoherent | hipHostMallocMapped));
    hipError_t e = hipExtMallocWithFlags((void**)&dfine, n * 8, hipDeviceMallocFinegrained);
    std::printf("hipExtMallocWithFlags(fine-grained): %s\n", hipGetErrorString(e));
    hipStream_t st; CK(hipStreamCreate(&st));
    using clk = std::chrono::steady_clock;
    double ref = 0; for (double v : src) ref += v;
    if (e == hipSuccess) {
        hipPointerAttribute_t a; CK(hipPointerGetAttributes(&a, dfine));
        std::printf("fine-grained: type %d hostPointer %p devicePointer %p\n", (int)a.type, a.hostPointer, a.devicePointer);
        // host write straight into it (if the mapping is not host-accessible this faults: run this experiment alone)
        for (int rep = 0; rep < 5; ++rep) {
            auto t0 = clk::now();
            std::memcpy(dfine, src.data(), n * 8);
            auto t1 = clk::now();
            hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, st, (const double*)dfine, n, dout);
            CK(hipStreamSynchronize(st));
            auto t2 = clk::now();
            double got = 0; CK(hipMemcpy(&got, dout, 8, hipMemcpyDeviceToHost));
            std::printf("BAR write: memcpy %.1f us, kernel+sync %.1f us, sum %s\n", std::chrono::duration<double, std::micro>(t1 - t0).count(),
                        std::chrono::duration<double, std::micro>(t2 - t1).count(), got == ref ? "ok" : "WRONG");
            src[rep] += 1.0; ref += 1.0;
        }
    }
    for (int rep = 0; rep < 5; ++rep) {
        auto t0 = clk::now();
        std::memcpy(hpin, src.data(), n * 8);
        auto t1 = clk::now();
        hipLaunchKernelGGL(k_pull, dim3(32), dim3(256), 0, st, (const double*)hpin, dplain, n);
        hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, st, (const double*)dplain, n, dout);
        CK(hipStreamSynchronize(st));
        auto t2 = clk::now();
        double got = 0; CK(hipMemcpy(&got, dout, 8, hipMemcpyDeviceToHost));
        std::printf("pinned + pull: memcpy %.1f us, kernels+sync %.1f us, sum %s\n", std::chrono::duration<double, std::micro>(t1 - t0).count(),
                    std::chrono::duration<double, std::micro>(t2 - t1).count(), got == ref ? "ok" : "WRONG");
        src[rep] += 1.0; ref += 1.0;
    }
    return 0;
}
