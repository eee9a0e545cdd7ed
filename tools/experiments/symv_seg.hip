// symv_seg.hip -- wider column segments for the lower-triangle GEMV tile (more contiguous bytes per row): timing only.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include "../../ellalgo-rs_amd/csrc/ell_kernels.hpp"
using namespace ellhip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int RW, int SEG, int WPE>
__global__ __launch_bounds__(256, WPE) void k_plain(const double* __restrict__ Q, long long ld, long long n, const double* __restrict__ g,
                                                    double* __restrict__ rowpart, double* __restrict__ colpart) {
    __shared__ double red[4][SYMV_H];
    symv_tile<RW, true, 0, SEG, false>(Q, ld, n, 0, n, g, rowpart, colpart, (long long)gridDim.x - 1 - blockIdx.x, (long long)blockIdx.y, red);
}
int main() {
    const long long n = 16384, ld = n + 16;
    double *Q, *g, *rp, *cp;
    CK(hipMalloc(&Q, (size_t)n * ld * 8)); CK(hipMemset(Q, 0, (size_t)n * ld * 8));
    CK(hipMalloc(&g, n * 8)); CK(hipMemset(g, 0, n * 8));
    CK(hipMalloc(&rp, (size_t)32 * n * 8)); CK(hipMalloc(&cp, (size_t)256 * n * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 5; ++i) launch();
        CK(hipDeviceSynchronize());
        float best = 1e9f, sum = 0;
        for (int i = 0; i < 20; ++i) {
            CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms); sum += ms;
        }
        printf("%-40s avg %.1f us  best %.1f us  (%.0f GB/s avg)\n", name, sum / 20 * 1e3, best * 1e3, 4.0 * n * n / (sum / 20 * 1e-3) / 1e9);
    };
    const unsigned ns = (unsigned)(n / SYMV_H);
#define GO(RW, SEG, WPE) timeit("64 x " #SEG ", " #RW " rows in flight, " #WPE " waves/SIMD", [&]() { hipLaunchKernelGGL((k_plain<RW, SEG, WPE>), dim3(ns, (unsigned)((n + SEG - 1) / SEG)), dim3(256), 0, 0, Q, ld, n, g, rp, cp); })
    for (int rep = 0; rep < 2; ++rep) {
        GO(2, 2048, 5);
        GO(1, 4096, 4);
        GO(2, 4096, 3);
        GO(1, 8192, 2);
        GO(1, 2048, 6);
        GO(4, 1024, 5);
        GO(2, 1024, 6);
    }
    return 0;
}
