// apply_depth.hip -- how long does the lower-triangle apply pass take for 32 recorded updates?  (k_apply_lower's
// coefficient reads made volatile so that the compiler does not preload NP x RG of them: 228 VGPRs at NP = 32, RG = 2)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../ellalgo-rs_amd/csrc/ell_kernels.hpp"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
namespace ellhip {
template <int NP, bool NT, int APL_RG = (NP == 16 ? 4 : 8)>
__global__ __launch_bounds__(256, 2) void k_apply_lower2(double* __restrict__ Q, long long ld, long long n,
                                                     long long nrows, long long row0,
                                                     const double* __restrict__ pend,
                                                     const double* __restrict__ cpend,
                                                     const DevState* __restrict__ st) {
    __shared__ double coef[NP][APL_TR];
    (void)st;
    const long long tile = (long long)gridDim.x - 1 - blockIdx.x;  // last (longest) rows first
    const long long lr0 = tile * APL_TR;                            // first local row of the tile
    if (lr0 >= nrows) return;
    const int nr = (int)((nrows - lr0 < APL_TR) ? nrows - lr0 : APL_TR);
    for (int idx = threadIdx.x; idx < NP * APL_TR; idx += 256) {
        const int j = idx / APL_TR, r = idx - j * APL_TR;
        coef[j][r] = (r < nr) ? cpend[j] * pend[(long long)j * n + row0 + lr0 + r] : 0.0;  // r_qg of src/ell.rs:119
    }
    __syncthreads();
    const long long gmax = row0 + lr0 + nr - 1;  // last global row of the tile
    long long cend = (gmax / 2 + 1) * 2;         // first column past the tile's diagonal, pair aligned
    if (cend > n) cend = n;
    double* base = Q + lr0 * ld;
    for (long long c = 2 * (long long)threadIdx.x; c < cend; c += 512) {
        double2_t vj[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) vj[j] = *reinterpret_cast<const double2_t*>(pend + (long long)j * n + c);
#pragma unroll 1  // (fully unrolled, the row groups kept 255 VGPRs + 68 AGPRs alive: occupancy 1)
        for (int r0 = 0; r0 < APL_TR; r0 += APL_RG) {
            double2_t x[APL_RG];
            bool on[APL_RG];
#pragma unroll
            for (int u = 0; u < APL_RG; ++u) {
                const int r = r0 + u;
                // a row takes part while the pair starts at or left of its diagonal (keeps the traffic at the trapezoid)
                on[u] = r < nr && c <= row0 + lr0 + r;
                if (on[u]) x[u] = ld_stream<NT, double2_t>(base + (long long)r * ld + c);
            }
#pragma unroll
            for (int j = 0; j < NP; ++j) {
#pragma unroll
                for (int u = 0; u < APL_RG; ++u) {  // per element still j ascending: the reference's order of roundings
                    const double cf = *reinterpret_cast<const volatile double*>(&coef[j][r0 + u]);
                    x[u].x = x[u].x - cf * vj[j].x;
                    x[u].y = x[u].y - cf * vj[j].y;
                }
            }
#pragma unroll
            for (int u = 0; u < APL_RG; ++u)
                if (on[u]) *reinterpret_cast<double2_t*>(base + (long long)(r0 + u) * ld + c) = x[u];
        }
    }
}


}
using namespace ellhip;
int main() {
    const long long n = 16384, ld = n + 16;
    double *Q, *pend, *cpend; DevState* st;
    CK(hipMalloc(&Q, (size_t)n * ld * 8)); CK(hipMemset(Q, 0, (size_t)n * ld * 8));
    CK(hipMalloc(&pend, (size_t)32 * n * 8)); CK(hipMemset(pend, 0, (size_t)32 * n * 8));
    CK(hipMalloc(&cpend, 32 * 8)); CK(hipMemset(cpend, 0, 32 * 8));
    CK(hipMalloc(&st, sizeof(DevState))); CK(hipMemset(st, 0, sizeof(DevState)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((n + APL_TR - 1) / APL_TR);
    auto timeit = [&](const char* name, int np, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        float best = 1e9f, sum = 0;
        for (int i = 0; i < 10; ++i) {
            CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms); sum += ms;
        }
        printf("%-44s avg %.1f us  best %.1f us  = %.1f us per recorded update (%.0f GB/s)\n", name, sum / 10 * 1e3, best * 1e3, sum / 10 * 1e3 / np, 8.0 * n * n / (sum / 10 * 1e-3) / 1e9);
    };
    timeit("k_apply_lower<16, nt, RG 4> (production)", 16, [&]() { hipLaunchKernelGGL((k_apply_lower<16, true, 4>), dim3(grid), dim3(256), 0, 0, Q, ld, n, n, 0LL, (const double*)pend, (const double*)cpend, (const DevState*)st); });
    timeit("volatile coefs <16, nt, RG 4>", 16, [&]() { hipLaunchKernelGGL((k_apply_lower2<16, true, 4>), dim3(grid), dim3(256), 0, 0, Q, ld, n, n, 0LL, (const double*)pend, (const double*)cpend, (const DevState*)st); });
    timeit("volatile coefs <32, nt, RG 2>", 32, [&]() { hipLaunchKernelGGL((k_apply_lower2<32, true, 2>), dim3(grid), dim3(256), 0, 0, Q, ld, n, n, 0LL, (const double*)pend, (const double*)cpend, (const DevState*)st); });
    timeit("volatile coefs <32, nt, RG 1>", 32, [&]() { hipLaunchKernelGGL((k_apply_lower2<32, true, 1>), dim3(grid), dim3(256), 0, 0, Q, ld, n, n, 0LL, (const double*)pend, (const double*)cpend, (const DevState*)st); });
    timeit("volatile coefs <24, nt, RG 2>", 24, [&]() { hipLaunchKernelGGL((k_apply_lower2<24, true, 2>), dim3(grid), dim3(256), 0, 0, Q, ld, n, n, 0LL, (const double*)pend, (const double*)cpend, (const DevState*)st); });
    timeit("volatile coefs <24, nt, RG 4>", 24, [&]() { hipLaunchKernelGGL((k_apply_lower2<24, true, 4>), dim3(grid), dim3(256), 0, 0, Q, ld, n, n, 0LL, (const double*)pend, (const double*)cpend, (const DevState*)st); });
    return 0;
}
