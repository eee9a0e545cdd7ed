// symv_timeline.hip -- per-workgroup start / end stamps of the production k_symv tile (symv_tile of ell_kernels.hpp) at
// n = 16384: when do the tiles of the static (strip, segment) grid end, and on which XCD / CU do the late ones run?
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include "../../ellalgo-rs_amd/csrc/ell_kernels.hpp"
using namespace ellhip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int RW, int SEG>
__global__ __launch_bounds__(256) void k_symv_stamped(const double* __restrict__ Q, long long ld, long long n,
                                                      const double* __restrict__ g, double* __restrict__ rowpart,
                                                      double* __restrict__ colpart, unsigned long long* stamps) {
    __shared__ double red[4][SYMV_H];
    const unsigned long long t0 = wall_clock64();
    const bool did = symv_tile<RW, true, 0, SEG, false>(Q, ld, n, 0, n, g, rowpart, colpart, (long long)gridDim.x - 1 - blockIdx.x,
                                                        (long long)blockIdx.y, red);
    if (threadIdx.x == 0) {
        const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
        unsigned xcc = 0, hwid = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        stamps[4 * lin] = t0;
        stamps[4 * lin + 1] = wall_clock64();
        stamps[4 * lin + 2] = did ? 1 : 0;
        stamps[4 * lin + 3] = ((unsigned long long)(xcc & 0xf) << 32) | hwid;
    }
}

int main(int argc, char** argv) {
    const long long n = 16384, ld = n + 16;
    double *Q, *g, *rowpart, *colpart;
    CK(hipMalloc(&Q, (size_t)n * ld * 8)); CK(hipMemset(Q, 0, (size_t)n * ld * 8));
    CK(hipMalloc(&g, n * 8)); CK(hipMemset(g, 0, n * 8));
    CK(hipMalloc(&rowpart, (size_t)8 * n * 8)); CK(hipMalloc(&colpart, (size_t)256 * n * 8));
    const unsigned nstrips = 256, nsegs = 8, total = nstrips * nsegs;
    unsigned long long* stamps; CK(hipMalloc(&stamps, (size_t)4 * total * 8));
    std::vector<unsigned long long> h((size_t)4 * total);
    for (int rep = 0; rep < 4; ++rep) {
        hipLaunchKernelGGL((k_symv_stamped<2, 2048>), dim3(nstrips, nsegs), dim3(256), 0, 0, Q, ld, n, g, rowpart, colpart, stamps);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (unsigned i = 0; i < total; ++i) if (h[4 * i + 2]) t0 = std::min(t0, h[4 * i]);
    std::vector<double> ends, starts, durs;
    std::map<unsigned, std::vector<double>> by_xcc;
    std::map<unsigned long long, int> per_cu;
    for (unsigned i = 0; i < total; ++i) if (h[4 * i + 2]) {
        const double e = (h[4 * i + 1] - t0) / 100.0, s = (h[4 * i] - t0) / 100.0;
        ends.push_back(e); starts.push_back(s); durs.push_back(e - s);
        const unsigned xcc = (unsigned)(h[4 * i + 3] >> 32);
        by_xcc[xcc].push_back(e);
        // HW_ID: cu_id bits 8..11, sh_id bit 12, se_id bits 13..15 (gfx9 layout)
        const unsigned hw = (unsigned)h[4 * i + 3];
        per_cu[((unsigned long long)xcc << 16) | ((hw >> 8) & 0xff)] += 1;
    }
    auto pct = [](std::vector<double> a, double f) { std::sort(a.begin(), a.end()); return a[(size_t)(f * (a.size() - 1))]; };
    printf("active tiles %zu; starts: p50 %.1f max %.1f us; ends: min %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f us; tile duration p10 %.1f p50 %.1f p90 %.1f max %.1f us\n",
           ends.size(), pct(starts, .5), pct(starts, 1), pct(ends, 0), pct(ends, .1), pct(ends, .5), pct(ends, .9), pct(ends, .99), pct(ends, 1),
           pct(durs, .1), pct(durs, .5), pct(durs, .9), pct(durs, 1));
    for (auto& kv : by_xcc) printf("  XCC %u: %zu tiles, ends p50 %.1f p90 %.1f max %.1f\n", kv.first, kv.second.size(), pct(kv.second, .5), pct(kv.second, .9), pct(kv.second, 1));
    std::map<int, int> hist;
    for (auto& kv : per_cu) hist[kv.second] += 1;
    printf("  tiles per (XCC, CU id): ");
    for (auto& kv : hist) printf("%d CUs hold %d; ", kv.second, kv.first);
    printf("\n  late tiles (end > p90): ");
    const double p90 = pct(ends, .9);
    int shown = 0;
    for (unsigned i = 0; i < total && shown < 24; ++i) if (h[4 * i + 2]) {
        const double e = (h[4 * i + 1] - t0) / 100.0;
        if (e > p90) { printf("(I=%u J=%u xcc=%u %.0f) ", nstrips - 1 - (i % nstrips), i / nstrips, (unsigned)(h[4 * i + 3] >> 32), e); ++shown; }
    }
    printf("\n");
    return 0;
}
