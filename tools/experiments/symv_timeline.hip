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

// order == nullptr: the production 2-D grid (strip = gridDim.x - 1 - blockIdx.x, segment = blockIdx.y);
// otherwise a 1-D grid over the ACTIVE tiles only, order[lin] = (strip << 8) | segment.
template <int RW, int SEG>
__global__ __launch_bounds__(256) void k_symv_stamped(const double* __restrict__ Q, long long ld, long long n,
                                                      const double* __restrict__ g, double* __restrict__ rowpart,
                                                      double* __restrict__ colpart, unsigned long long* stamps,
                                                      const unsigned* __restrict__ order) {
    __shared__ double red[4][SYMV_H];
    const unsigned long long t0 = wall_clock64();
    long long I = (long long)gridDim.x - 1 - blockIdx.x, J = blockIdx.y;
    if (order) { const unsigned o = order[blockIdx.x]; I = o >> 8; J = o & 0xff; }
    const bool did = symv_tile<RW, true, 0, SEG, false>(Q, ld, n, 0, n, g, rowpart, colpart, I, J, red);
    if (threadIdx.x == 0) {
        const unsigned lin = order ? blockIdx.x : blockIdx.y * gridDim.x + blockIdx.x;
        unsigned xcc = 0, hwid = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        stamps[4 * lin] = t0;
        stamps[4 * lin + 1] = wall_clock64();
        stamps[4 * lin + 2] = did ? 1 : 0;
        stamps[4 * lin + 3] = ((unsigned long long)(xcc & 0xf) << 32) | hwid;
    }
}


static std::vector<unsigned> make_order(int mode, unsigned nstrips, unsigned spseg /* strips per segment */) {
    // full tiles (J < I / spseg) and diagonal tiles (J == I / spseg, (I % spseg + 1) / spseg of a full tile)
    std::vector<unsigned> full, diag;
    for (unsigned J = 0; J * spseg < nstrips; ++J)
        for (unsigned I = nstrips; I-- > 0;) if (J < I / spseg) full.push_back((I << 8) | J);
    for (unsigned I = nstrips; I-- > 0;) diag.push_back((I << 8) | (I / spseg));
    std::stable_sort(diag.begin(), diag.end(), [&](unsigned a, unsigned b) { return ((a >> 8) % spseg) > ((b >> 8) % spseg); });
    std::vector<unsigned> o;
    if (mode == 1) {           // every full tile, then the diagonal tiles from the largest to the smallest
        o = full; o.insert(o.end(), diag.begin(), diag.end());
    } else if (mode == 2) {    // diagonal tiles first (largest first), then the full ones
        o = diag; o.insert(o.end(), full.begin(), full.end());
    } else if (mode == 3) {    // dealt by hand for "workgroup lin lands on CU lin % 256": 3 full + big diag + small diag, or 4 full
        const size_t nd = diag.size(), half = nd / 2;
        o.insert(o.end(), full.begin(), full.begin() + 768);
        o.insert(o.end(), diag.begin(), diag.begin() + half);                 // CUs 0..127: the big diagonal tiles
        o.insert(o.end(), full.begin() + 768, full.end());                    // CUs 128..255: a fourth full tile
        for (size_t k = 0; k < half; ++k) o.push_back(diag[nd - 1 - k]);      // CUs 0..127: smallest with biggest
    }
    return o;
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
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    unsigned* d_order = nullptr;
    std::vector<unsigned> order;
    if (mode > 0) {
        order = make_order(mode, nstrips, 2048 / 64);
        CK(hipMalloc(&d_order, order.size() * 4));
        CK(hipMemcpy(d_order, order.data(), order.size() * 4, hipMemcpyHostToDevice));
    }
    const dim3 grid = mode > 0 ? dim3((unsigned)order.size()) : dim3(nstrips, nsegs);
    CK(hipMemset(stamps, 0, (size_t)4 * total * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 5; ++rep)
        hipLaunchKernelGGL((k_symv_stamped<2, 2048>), grid, dim3(256), 0, 0, Q, ld, n, g, rowpart, colpart, stamps, d_order);
    CK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0;
    for (int rep = 0; rep < 20; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_symv_stamped<2, 2048>), grid, dim3(256), 0, 0, Q, ld, n, g, rowpart, colpart, stamps, d_order);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms); sum += ms;
    }
    printf("mode %d: %u workgroups; kernel avg %.1f us, best %.1f us\n", mode, grid.x * grid.y, sum / 20 * 1000, best * 1000);
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
        const unsigned I = mode > 0 ? order[i] >> 8 : nstrips - 1 - (i % nstrips), J = mode > 0 ? order[i] & 0xff : i / nstrips;
        if (e > p90) { printf("(lin=%u I=%u J=%u xcc=%u %.0f) ", i, I, J, (unsigned)(h[4 * i + 3] >> 32), e); ++shown; }
    }
    printf("\n");
    // how the dispatcher deals: does workgroup lin share its CU with lin + 256?  and the per-CU totals
    {
        auto cu_of = [&](unsigned i) { const unsigned hw = (unsigned)h[4 * i + 3]; return (unsigned)(((h[4 * i + 3] >> 32) << 16) | ((hw >> 8) & 0xff)); };
        const unsigned nw = mode > 0 ? (unsigned)order.size() : total;
        unsigned same = 0, cnt = 0;
        for (unsigned i = 0; i + 256 < nw; ++i) if (h[4 * i + 2] && h[4 * (i + 256) + 2]) { ++cnt; same += cu_of(i) == cu_of(i + 256); }
        printf("  lin and lin + 256 on the same CU: %u of %u\n", same, cnt);
        std::map<unsigned, double> cu_tiles, cu_end;
        for (unsigned i = 0; i < nw; ++i) if (h[4 * i + 2]) {
            const unsigned I = mode > 0 ? order[i] >> 8 : nstrips - 1 - (i % nstrips), J = mode > 0 ? order[i] & 0xff : i / nstrips;
            const double sz = (J < I / 32) ? 1.0 : ((I % 32) + 1) / 32.0;
            cu_tiles[cu_of(i)] += sz;
            cu_end[cu_of(i)] = std::max(cu_end[cu_of(i)], (h[4 * i + 1] - t0) / 100.0);
        }
        std::vector<double> lt, le;
        for (auto& kv : cu_tiles) { lt.push_back(kv.second); le.push_back(cu_end[kv.first]); }
        printf("  per CU: full-tile equivalents min %.2f p50 %.2f max %.2f; last end min %.1f p50 %.1f p90 %.1f max %.1f us\n",
               pct(lt, 0), pct(lt, .5), pct(lt, 1), pct(le, 0), pct(le, .5), pct(le, .9), pct(le, 1));
        printf("  first 24 workgroups -> (xcc, cu): ");
        for (unsigned i = 0; i < 24; ++i) printf("%x ", cu_of(i));
        printf("\n  workgroups 256..279      : ");
        for (unsigned i = 256; i < 280; ++i) printf("%x ", cu_of(i));
        printf("\n");
    }
    return 0;
}
