// symv_bal.hip -- prototype: the lower-triangle GEMV's tiles dealt to the CUs by BYTES.
// A CU streams ~27-34 GB/s however many tiles it holds (profiles/r02/symv_timeline_production_kernel.txt), and the
// dispatcher gives the CUs 3, 4 or 5 of the 1152 tiles at n = 16384, so the launch lasts as long as the CUs with 5 full
// tiles.  Here: a persistent grid of (CUs x occupancy) workgroups; a workgroup finds out which CU it runs on (XCC_ID +
// HW_ID, numbered densely in order of first arrival) and pulls tiles from THAT CU's list (host-built, longest-processing-
// time-first by loaded bytes); a workgroup whose list is empty steals from the others.  Same tiles, same outputs.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../ellalgo-rs_amd/csrc/ell_kernels.hpp"
using namespace ellhip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int BAL_MAXL = 512, BAL_MAXI = 32;
struct BalCtl {
    int cu_of[2048];            // (xcc << 8 | se/sh/cu bits) -> dense CU number, -1 unknown, -2 being assigned
    int ncu;
    unsigned exited;
    int nlist;
    int cnt[BAL_MAXL * 32];     // next item of list c, one 128-byte line each (c * 32): the start-up atomics of 1280
                                // workgroups on 8 shared lines cost ~50 us
    int len[BAL_MAXL];
    int items[BAL_MAXL][BAL_MAXI];  // strip << 8 | segment
};

template <int RW, int SEG, bool STAMP = false>
__global__ __launch_bounds__(256, RW == 2 ? 5 : 4) void k_symv_bal(const double* __restrict__ Q, long long ld, long long n,
                                                  const double* __restrict__ g, double* __restrict__ rowpart,
                                                  double* __restrict__ colpart, BalCtl* __restrict__ ctl,
                                                  unsigned long long* stamps) {
    __shared__ double red[4][SYMV_H];
    __shared__ int s_list, s_item;
    const int nlist = ctl->nlist;
    if (threadIdx.x == 0) {
        unsigned xcc = 0, hwid = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        const int raw = (int)(((xcc & 0x7) << 8) | ((hwid >> 8) & 0xff));
        int d = __hip_atomic_load(&ctl->cu_of[raw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d < 0) {
            if (atomicCAS(&ctl->cu_of[raw], -1, -2) == -1) {
                d = atomicAdd(&ctl->ncu, 1);
                __hip_atomic_store(&ctl->cu_of[raw], d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                for (int spin = 0; spin < (1 << 20); ++spin) {
                    d = __hip_atomic_load(&ctl->cu_of[raw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (d >= 0) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (d < 0) d = nlist;  // gave up: steal only
            }
        }
        s_list = d;
    }
    __syncthreads();
    const int mine = s_list;
    int victim = mine;  // the list this workgroup currently pulls from
    for (;;) {
        if (threadIdx.x == 0) {
            int item = -1;
            if (victim < nlist) {
                const int k = atomicAdd(&ctl->cnt[victim * 32], 1);
                if (k < ctl->len[victim]) item = ctl->items[victim][k];
            }
            s_item = item;
        }
        __syncthreads();
        int item = s_item;
        __syncthreads();
        if (item < 0) {
            // own (or current victim's) list is exhausted: wave 0 looks for a list with items left
            if (threadIdx.x < 64) {
                int found = -1;
                for (int base = 0; base < nlist && found < 0; base += 64) {
                    const int c = (((mine < nlist ? mine : 0) + 1 + base + (int)threadIdx.x) % nlist);
                    const bool has = (base + (int)threadIdx.x < nlist) &&
                                     __hip_atomic_load(&ctl->cnt[c * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ctl->len[c];
                    const unsigned long long m = __ballot(has);
                    if (m) found = (((mine < nlist ? mine : 0) + 1 + base + (int)__builtin_ctzll(m)) % nlist);
                }
                if (threadIdx.x == 0) s_list = found;
            }
            __syncthreads();
            victim = s_list;
            __syncthreads();
            if (victim < 0) break;
            continue;
        }
        unsigned long long t0 = 0;
        if (STAMP && threadIdx.x == 0) t0 = wall_clock64();
        symv_tile<RW, true, 0, SEG, false>(Q, ld, n, 0, n, g, rowpart, colpart, (long long)(item >> 8), (long long)(item & 0xff), red);
        if (STAMP && threadIdx.x == 0) {
            const int slot = (item >> 8) * 16 + (item & 0xff);
            stamps[3 * slot] = t0; stamps[3 * slot + 1] = wall_clock64(); stamps[3 * slot + 2] = (unsigned long long)(mine < nlist ? mine : 999);
        }
        __syncthreads();  // red is reused
    }
    if (threadIdx.x == 0) {
        const unsigned e = atomicAdd(&ctl->exited, 1u);
        if (e == gridDim.x - 1) {  // last one out re-arms the lists
            for (int c = 0; c < nlist; ++c) __hip_atomic_store(&ctl->cnt[c * 32], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl->exited, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int RW, int SEG>
__global__ __launch_bounds__(256) void k_symv_plain(const double* __restrict__ Q, long long ld, long long n,
                                                    const double* __restrict__ g, double* __restrict__ rowpart,
                                                    double* __restrict__ colpart) {
    __shared__ double red[4][SYMV_H];
    symv_tile<RW, true, 0, SEG, false>(Q, ld, n, 0, n, g, rowpart, colpart, (long long)gridDim.x - 1 - blockIdx.x, (long long)blockIdx.y, red);
}

int main(int argc, char** argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 16384;
    const double fixed = argc > 2 ? atof(argv[2]) : 0.05;   // per-tile fixed cost, in full tiles
    const long long ld = n + 16;
    constexpr int SEG = 2048;
    const int nstrips = (int)((n + SYMV_H - 1) / SYMV_H), nsegs = (int)((n + SEG - 1) / SEG);
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_symv_bal<2, SEG>, 256, 0));
    printf("n = %lld: %d strips x %d segments, %d CUs, occupancy %d\n", n, nstrips, nsegs, ncu, occ);
    // host data
    std::vector<double> hQ((size_t)n * ld), hg(n);
    unsigned long long x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (double)(x >> 11) / 9007199254740992.0 - 0.5; };
    for (auto& v : hQ) v = rnd();
    for (auto& v : hg) v = rnd();
    double *Q, *g, *rp, *cp, *rp2, *cp2;
    CK(hipMalloc(&Q, hQ.size() * 8)); CK(hipMemcpy(Q, hQ.data(), hQ.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&g, n * 8)); CK(hipMemcpy(g, hg.data(), n * 8, hipMemcpyHostToDevice));
    const size_t nrp = (size_t)nsegs * n, ncp = (size_t)nstrips * n;
    CK(hipMalloc(&rp, nrp * 8)); CK(hipMalloc(&cp, ncp * 8)); CK(hipMalloc(&rp2, nrp * 8)); CK(hipMalloc(&cp2, ncp * 8));
    CK(hipMemset(rp, 0, nrp * 8)); CK(hipMemset(cp, 0, ncp * 8)); CK(hipMemset(rp2, 0, nrp * 8)); CK(hipMemset(cp2, 0, ncp * 8));
    // tiles and their cost (loaded bytes in full-tile units + a fixed part), LPT onto the CUs
    struct T { int item; double cost; };
    std::vector<T> tiles;
    for (int J = 0; J < nsegs; ++J)
        for (int I = 0; I < nstrips; ++I) {
            const long long r0 = (long long)I * SYMV_H, c0 = (long long)J * SEG;
            if (c0 > r0 + SYMV_H - 1) continue;
            double el = 0;
            for (long long r = r0; r < std::min<long long>(r0 + SYMV_H, n); ++r) el += (double)std::max<long long>(0, std::min<long long>(SEG, r - c0 + 2));
            tiles.push_back({(I << 8) | J, el / (64.0 * SEG) + fixed});
        }
    const int listmode = argc > 3 ? atoi(argv[3]) : 0;
    static BalCtl h;
    memset(&h, 0, sizeof(h));
    for (auto& v : h.cu_of) v = -1;
    h.nlist = ncu;
    std::vector<double> load(ncu, 0.0);
    auto lpt = [&](std::vector<T> v) {
        std::sort(v.begin(), v.end(), [](const T& a, const T& b) { return a.cost > b.cost; });
        for (auto& t : v) {
            int best = 0;
            for (int c = 1; c < ncu; ++c) if (load[c] < load[best]) best = c;
            if (h.len[best] >= BAL_MAXI) { printf("list overflow\n"); exit(1); }
            h.items[best][h.len[best]++] = t.item;
            load[best] += t.cost;
        }
    };
    if (listmode == 0) {
        lpt(tiles);
    } else {
        // locality first: the FULL tiles in (strip descending, segment ascending) order, dealt in runs -- a CU gets
        // consecutive segments of one strip (the same 64 rows: the same pages) -- then the diagonal tiles by LPT
        std::vector<T> full, diag;
        for (auto& t : tiles) (t.cost >= 1.0 + fixed - 1e-9 ? full : diag).push_back(t);
        std::sort(full.begin(), full.end(), [](const T& a, const T& b) { const int ia = a.item >> 8, ib = b.item >> 8; return ia != ib ? ia > ib : (a.item & 0xff) < (b.item & 0xff); });
        const double per = (double)full.size() / ncu;
        for (size_t k = 0; k < full.size(); ++k) {
            const int c = std::min(ncu - 1, (int)(k / per));
            h.items[c][h.len[c]++] = full[k].item;
            load[c] += full[k].cost;
        }
        lpt(diag);
    }
    printf("%zu tiles; per-CU load min %.3f max %.3f full tiles, list lengths %d..%d\n", tiles.size(), *std::min_element(load.begin(), load.end()),
           *std::max_element(load.begin(), load.end()), *std::min_element(h.len, h.len + ncu), *std::max_element(h.len, h.len + ncu));
    BalCtl* ctl; CK(hipMalloc(&ctl, sizeof(BalCtl))); CK(hipMemcpy(ctl, &h, sizeof(BalCtl), hipMemcpyHostToDevice));
    unsigned long long* stamps; CK(hipMalloc(&stamps, (size_t)3 * nstrips * 16 * 8)); CK(hipMemset(stamps, 0, (size_t)3 * nstrips * 16 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 5; ++i) launch();
        CK(hipDeviceSynchronize());
        float best = 1e9f, sum = 0;
        for (int i = 0; i < 20; ++i) {
            CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms); sum += ms;
        }
        printf("%-52s avg %.1f us  best %.1f us  (%.0f GB/s avg)\n", name, sum / 20 * 1e3, best * 1e3, 4.0 * n * n / (sum / 20 * 1e-3) / 1e9);
    };
    for (int rep = 0; rep < 1; ++rep) {
        timeit("k_symv static 2-D grid (production)", [&]() { hipLaunchKernelGGL((k_symv_plain<2, SEG>), dim3(nstrips, nsegs), dim3(256), 0, 0, Q, ld, n, g, rp, cp); });
        for (int w : {occ + 1, occ, occ - 1, occ - 2})
            if (w >= 2) {
                char name[96]; snprintf(name, sizeof(name), "k_symv_bal RW2, %d workgroups per CU", w);
                timeit(name, [&]() { hipLaunchKernelGGL((k_symv_bal<2, SEG>), dim3(ncu * w), dim3(256), 0, 0, Q, ld, n, g, rp2, cp2, ctl, (unsigned long long*)nullptr); });
            }
        timeit("k_symv static 2-D grid RW4", [&]() { hipLaunchKernelGGL((k_symv_plain<4, SEG>), dim3(nstrips, nsegs), dim3(256), 0, 0, Q, ld, n, g, rp, cp); });
        for (int w : {4, 3, 2}) {
            char name[96]; snprintf(name, sizeof(name), "k_symv_bal RW4, %d workgroups per CU", w);
            timeit(name, [&]() { hipLaunchKernelGGL((k_symv_bal<4, SEG>), dim3(ncu * w), dim3(256), 0, 0, Q, ld, n, g, rp2, cp2, ctl, (unsigned long long*)nullptr); });
        }
        for (int w : {3, 2}) {
            char name[96]; snprintf(name, sizeof(name), "k_symv_bal RW8, %d workgroups per CU", w);
            timeit(name, [&]() { hipLaunchKernelGGL((k_symv_bal<8, SEG>), dim3(ncu * w), dim3(256), 0, 0, Q, ld, n, g, rp2, cp2, ctl, (unsigned long long*)nullptr); });
        }
    }
    // same outputs?
    CK(hipMemset(rp2, 0, nrp * 8)); CK(hipMemset(cp2, 0, ncp * 8));
    hipLaunchKernelGGL((k_symv_bal<2, SEG, true>), dim3(ncu * occ), dim3(256), 0, 0, Q, ld, n, g, rp2, cp2, ctl, stamps);
    CK(hipDeviceSynchronize());
    std::vector<double> a(nrp), b(nrp), c(ncp), d(ncp);
    CK(hipMemcpy(a.data(), rp, nrp * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), rp2, nrp * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(c.data(), cp, ncp * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(d.data(), cp2, ncp * 8, hipMemcpyDeviceToHost));
    printf("row partials %s, column partials %s\n", memcmp(a.data(), b.data(), nrp * 8) ? "DIFFER" : "identical", memcmp(c.data(), d.data(), ncp * 8) ? "DIFFER" : "identical");
    BalCtl* back = new BalCtl; CK(hipMemcpy(back, ctl, sizeof(BalCtl), hipMemcpyDeviceToHost));
    printf("CUs numbered: %d\n", back->ncu);
    // per-CU end times of the stamped launch
    std::vector<unsigned long long> hs((size_t)3 * nstrips * 16); CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull; for (size_t i = 0; i < hs.size(); i += 3) if (hs[i]) t0 = std::min(t0, hs[i]);
    std::vector<double> cuend(1000, 0.0), ends; int stolen = 0;
    for (size_t i = 0; i < hs.size(); i += 3) if (hs[i]) { const double e = (hs[i + 1] - t0) / 100.0; ends.push_back(e); cuend[hs[i + 2]] = std::max(cuend[hs[i + 2]], e); }
    std::sort(ends.begin(), ends.end());
    std::vector<double> ce; for (int cidx = 0; cidx < ncu; ++cidx) if (cuend[cidx] > 0) ce.push_back(cuend[cidx]);
    std::sort(ce.begin(), ce.end());
    printf("tile ends: p10 %.1f p50 %.1f p90 %.1f max %.1f us; last tile end per CU: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f us (%zu CUs)\n",
           ends[ends.size() / 10], ends[ends.size() / 2], ends[ends.size() * 9 / 10], ends.back(), ce.front(), ce[ce.size() / 10], ce[ce.size() / 2], ce[ce.size() * 9 / 10], ce.back(), ce.size());
    (void)stolen;
    return 0;
}
