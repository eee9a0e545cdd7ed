// symm_glds.hip -- round-4 experiment: k_symm_glds (LDS-DMA ring, symm_glds_kernel.hpp) against k_symm_mfma: time per pass
// and the partial sums compared bit for bit.   Usage: symm_glds [n] [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "symm_lc_kernel.hpp"

using namespace ellhip;

#define CK(x)                                                      \
    do {                                                           \
        hipError_t e = (x);                                        \
        if (e != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); \
            exit(1);                                               \
        }                                                          \
    } while (0)

__global__ void k_fill_sym(double* Q, long long ld, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n * ld; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / ld, c = i - r * ld;
        if (c >= n) { Q[i] = 0.0; continue; }
        unsigned long long h = ((unsigned long long)r * 0x9E3779B97F4A7C15ull) ^ ((unsigned long long)c * 0xBF58476D1CE4E5B9ull);
        h ^= h >> 29; h *= 0x94D049BB133111EBull; h ^= h >> 32;
        // NOT symmetric on purpose: the kernels may read the lower triangle only (what lies above the diagonal is stale)
        Q[i] = (double)(h & 0xFFFFFFFFFFFFFull) / 4503599627370496.0 - 0.5 + (r == c ? 2.0 : 0.0);
    }
}
__global__ void k_fill_vec(double* g, long long m) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x) {
        unsigned long long h = (unsigned long long)(i + 12345) * 0xD6E8FEB86659FD93ull;
        h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32;
        g[i] = ((double)(h & 0xFFFFFFFFFFFFFull) / 4503599627370496.0 - 0.5) * 0.7390851332151607;
    }
}

static long long n, ld;
static int rounds;
static hipEvent_t e0, e1;

template <typename F>
static double timeit(const char* name, F&& fn) {
    std::vector<float> ms;
    for (int r = 0; r < rounds + 2; ++r) {
        CK(hipEventRecord(e0, 0));
        fn();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (r >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double med = ms[ms.size() / 2];
    printf("%-44s med %.4f ms  min %.4f   %.0f GB/s of 4n^2\n", name, med, ms[0], 4.0 * n * n / 1e9 / (med * 1e-3));
    CK(hipGetLastError());
    return med;
}

template <int SEG, int D, int MODE = 0, bool STAMP = false>
static void launch_glds(const double* Q, long long row0, long long nrows, const double* gT, int lv, double* rp, double* cp,
                        long long rs, long long cs, const DevState* st) {
    static bool once = false;
    const size_t lds = (size_t)4 * D * SGL_SLOT * sizeof(double);
    if (!once) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_symm_glds<SEG, D, MODE, STAMP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        once = true;
    }
    const unsigned nstrips = (unsigned)((nrows + SYMV_H - 1) / SYMV_H), nsegs = (unsigned)((n + SEG - 1) / SEG);
    hipLaunchKernelGGL((k_symm_glds<SEG, D, MODE, STAMP>), dim3(nstrips, nsegs), dim3(MODE >= 6 ? 512 : 256), lds, 0, Q, ld, n, row0, nrows, gT, lv, rp, cp, rs, cs, st);
}

int main(int argc, char** argv) {
    n = argc > 1 ? atoll(argv[1]) : 16384;
    rounds = argc > 2 ? atoi(argv[2]) : 10;
    if (n % 64) { fprintf(stderr, "n must be a multiple of 64\n"); return 1; }
    ld = n + 16;
    constexpr int SEG = 2048;
    const long long nstrips = n / SYMV_H, nsegs = (n + SEG - 1) / SEG;
    const long long rs = nsegs * n, cs = nstrips * n;
    const int NV = 16;
    double *Q, *g, *gT, *rp[2], *cp[2];
    DevState* st;
    CK(hipMalloc(&Q, (size_t)n * ld * 8));
    CK(hipMalloc(&g, (size_t)NV * n * 8));
    CK(hipMalloc(&gT, (size_t)NV * n * 8));
    for (int k = 0; k < 2; ++k) {
        CK(hipMalloc(&rp[k], (size_t)NV * rs * 8 * 8));
        CK(hipMalloc(&cp[k], (size_t)NV * cs * 8));
    }
    CK(hipMalloc(&st, sizeof(DevState)));
    CK(hipMemset(st, 0, sizeof(DevState)));
    hipLaunchKernelGGL(k_fill_sym, dim3(4096), dim3(256), 0, 0, Q, ld, n);
    hipLaunchKernelGGL(k_fill_vec, dim3(256), dim3(256), 0, 0, g, (long long)NV * n);
    CK(hipDeviceSynchronize());
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("n=%lld ld=%lld rounds=%d   4n^2 = %.1f MB\n", n, ld, rounds, 4.0 * n * n / 1e6);
    std::vector<double> a((size_t)NV * std::max(rs, cs)), b((size_t)NV * std::max(rs, cs));
    auto same = [&](const char* what) {
        bool ok = true;
        CK(hipMemcpy(a.data(), rp[0], (size_t)NV * rs * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), rp[1], (size_t)NV * rs * 8, hipMemcpyDeviceToHost));
        ok = ok && memcmp(a.data(), b.data(), (size_t)NV * rs * 8) == 0;
        CK(hipMemcpy(a.data(), cp[0], (size_t)NV * cs * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), cp[1], (size_t)NV * cs * 8, hipMemcpyDeviceToHost));
        ok = ok && memcmp(a.data(), b.data(), (size_t)NV * cs * 8) == 0;
        printf("   %-40s partial sums vs k_symm_mfma: %s\n", what, ok ? "bit-identical" : "DIFFERENT");
        return ok;
    };
    struct Case { long long row0, nrows; int lv; };
    const Case cases[] = {{0, n, 16}, {0, n, 3}, {n / 4, n / 2, 16}, {n - 64, 64, 9}};
    bool all = true;
    for (const Case& c : cases) {
        const unsigned ns_ = (unsigned)(c.nrows / SYMV_H);
        hipLaunchKernelGGL(k_pack_grads, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, 0, (const double*)g, n, c.lv, n, gT);
        for (int k = 0; k < 2; ++k) {
            CK(hipMemset(rp[k], 0, (size_t)NV * rs * 8));
            CK(hipMemset(cp[k], 0, (size_t)NV * cs * 8));
        }
        char nm[96];
        snprintf(nm, sizeof nm, "k_symm_mfma rows [%lld, +%lld) lv %d", c.row0, c.nrows, c.lv);
        timeit(nm, [&] {
            hipLaunchKernelGGL((k_symm_mfma<true, SEG>), dim3(ns_, (unsigned)nsegs), dim3(256), 0, 0, (const double*)Q + c.row0 * ld, ld, n,
                               c.row0, c.nrows, (const double*)gT, c.lv, rp[0], cp[0], rs, cs, (const DevState*)st);
        });
#define VARIANT(D)                                                                                                              \
    {                                                                                                                           \
        CK(hipMemset(rp[1], 0, (size_t)NV * rs * 8));                                                                           \
        CK(hipMemset(cp[1], 0, (size_t)NV * cs * 8));                                                                           \
        snprintf(nm, sizeof nm, "k_symm_glds<D=%d> rows [%lld, +%lld) lv %d", D, c.row0, c.nrows, c.lv);                        \
        timeit(nm, [&] { launch_glds<SEG, D>((const double*)Q + c.row0 * ld, c.row0, c.nrows, gT, c.lv, rp[1], cp[1], rs, cs, st); }); \
        CK(hipDeviceSynchronize());                                                                                             \
        all = same(nm) && all;                                                                                                  \
    }
        VARIANT(2)
        VARIANT(3)
#undef VARIANT
        snprintf(nm, sizeof nm, "loading skeleton D=2 (no MFMA)");
        timeit(nm, [&] { launch_glds<SEG, 2, 1>((const double*)Q + c.row0 * ld, c.row0, c.nrows, gT, c.lv, rp[1], cp[1], rs, cs, st); });
        snprintf(nm, sizeof nm, "MFMA only D=2 (no loads)");
        timeit(nm, [&] { launch_glds<SEG, 2, 2>((const double*)Q + c.row0 * ld, c.row0, c.nrows, gT, c.lv, rp[1], cp[1], rs, cs, st); });
        snprintf(nm, sizeof nm, "MFMA only D=3 (no loads)");
        timeit(nm, [&] { launch_glds<SEG, 3, 2>((const double*)Q + c.row0 * ld, c.row0, c.nrows, gT, c.lv, rp[1], cp[1], rs, cs, st); });
        snprintf(nm, sizeof nm, "loads never waited for D=2");
        timeit(nm, [&] { launch_glds<SEG, 2, 3>((const double*)Q + c.row0 * ld, c.row0, c.nrows, gT, c.lv, rp[1], cp[1], rs, cs, st); });
        snprintf(nm, sizeof nm, "loads never waited for D=3");
        timeit(nm, [&] { launch_glds<SEG, 3, 3>((const double*)Q + c.row0 * ld, c.row0, c.nrows, gT, c.lv, rp[1], cp[1], rs, cs, st); });
        snprintf(nm, sizeof nm, "loading skeleton D=3 (no MFMA)");
        timeit(nm, [&] { launch_glds<SEG, 3, 1>((const double*)Q + c.row0 * ld, c.row0, c.nrows, gT, c.lv, rp[1], cp[1], rs, cs, st); });
    }
    // ---- tile width: the pass lasts as long as its last workgroup, and a 64 x 2048 tile is 1 / 1028 of the work on 512 slots
    {
        hipLaunchKernelGGL(k_pack_grads, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, 0, (const double*)g, n, 16, n, gT);
        char nm[96];
#define SEGRUN(SEGW)                                                                                                               \
    {                                                                                                                              \
        const long long nsg = (n + SEGW - 1) / SEGW;                                                                               \
        if (nsg * n <= rs * 8) {                                                                                                   \
            snprintf(nm, sizeof nm, "k_symm_mfma<%d> lv 16", SEGW);                                                                \
            timeit(nm, [&] {                                                                                                       \
                hipLaunchKernelGGL((k_symm_mfma<true, SEGW>), dim3((unsigned)nstrips, (unsigned)nsg), dim3(256), 0, 0, (const double*)Q, ld, n, \
                                   0LL, n, (const double*)gT, 16, rp[0], cp[0], nsg * n, cs, (const DevState*)st);                 \
            });                                                                                                                    \
            snprintf(nm, sizeof nm, "k_symm_glds<%d, D=2> lv 16", SEGW);                                                           \
            timeit(nm, [&] {                                                                                                       \
                const size_t lds = (size_t)4 * 2 * SGL_SLOT * sizeof(double);                                                      \
                static bool once = false;                                                                                          \
                if (!once) {                                                                                                       \
                    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_symm_glds<SEGW, 2, 0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                    once = true;                                                                                                   \
                }                                                                                                                  \
                hipLaunchKernelGGL((k_symm_glds<SEGW, 2, 0, false>), dim3((unsigned)nstrips, (unsigned)nsg), dim3(256), lds, 0, (const double*)Q, ld, n, \
                                   0LL, n, (const double*)gT, 16, rp[1], cp[1], nsg * n, cs, (const DevState*)st);                 \
            });                                                                                                                    \
        }                                                                                                                          \
    }
        SEGRUN(2048)
        SEGRUN(1024)
        SEGRUN(512)
        SEGRUN(256)
#undef SEGRUN
    }
    // ---- tiles from a queue, largest first, 2 workgroups per CU (symm_queue_kernel.hpp)
    {
        hipLaunchKernelGGL(k_pack_grads, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, 0, (const double*)g, n, 16, n, gT);
        for (int k = 0; k < 2; ++k) {
            CK(hipMemset(rp[k], 0, (size_t)NV * rs * 8));
            CK(hipMemset(cp[k], 0, (size_t)NV * cs * 8));
        }
        hipLaunchKernelGGL((k_symm_mfma<true, SEG>), dim3((unsigned)nstrips, (unsigned)nsegs), dim3(256), 0, 0, (const double*)Q, ld, n, 0LL, n,
                           (const double*)gT, 16, rp[0], cp[0], rs, cs, (const DevState*)st);
        std::vector<SymmTile> tl;
        std::vector<int> nb;
        for (int I = 0; I < (int)nstrips; ++I)
            for (int J = 0; J < (int)nsegs; ++J) {
                const long long r0 = (long long)I * SYMV_H, c0 = (long long)J * SEG;
                if (c0 > r0 + SYMV_H - 1) continue;
                tl.push_back({I, J});
            }
        auto blocks_of = [&](const SymmTile& t) {
            const long long r0 = (long long)t.I * SYMV_H, c0 = (long long)t.J * SEG;
            const long long cend = std::min<long long>(c0 + SEG, r0 + SYMV_H);
            return (cend - c0) / 16;
        };
        std::stable_sort(tl.begin(), tl.end(), [&](const SymmTile& a, const SymmTile& b) { return blocks_of(a) > blocks_of(b); });
        SymmTile* d_tl;
        unsigned* d_cnt;
        CK(hipMalloc(&d_tl, tl.size() * sizeof(SymmTile)));
        CK(hipMalloc(&d_cnt, 4));
        CK(hipMemcpy(d_tl, tl.data(), tl.size() * sizeof(SymmTile), hipMemcpyHostToDevice));
        const int ntiles = (int)tl.size();
        char nm[96];
        for (int wgs : {256, 512, 768}) {
            snprintf(nm, sizeof nm, "k_symm_q_reg, %d tiles, %d workgroups", ntiles, wgs);
            timeit(nm, [&] {
                CK(hipMemsetAsync(d_cnt, 0, 4, 0));
                hipLaunchKernelGGL((k_symm_q_reg<true, SEG>), dim3((unsigned)wgs), dim3(256), 0, 0, (const double*)Q, ld, n, 0LL, n, (const double*)gT,
                                   16, rp[1], cp[1], rs, cs, (const DevState*)st, (const SymmTile*)d_tl, ntiles, d_cnt);
            });
            CK(hipDeviceSynchronize());
            all = same(nm) && all;
            CK(hipMemset(rp[1], 0, (size_t)NV * rs * 8));
            CK(hipMemset(cp[1], 0, (size_t)NV * cs * 8));
        }
        const size_t lds2 = (size_t)4 * 2 * SGL_SLOT * sizeof(double);
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_symm_q_glds<SEG, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        for (int wgs : {256, 512}) {
            snprintf(nm, sizeof nm, "k_symm_q_glds<D=2>, %d tiles, %d workgroups", ntiles, wgs);
            timeit(nm, [&] {
                CK(hipMemsetAsync(d_cnt, 0, 4, 0));
                hipLaunchKernelGGL((k_symm_q_glds<SEG, 2>), dim3((unsigned)wgs), dim3(256), lds2, 0, (const double*)Q, ld, n, 0LL, n, (const double*)gT,
                                   16, rp[1], cp[1], rs, cs, (const DevState*)st, (const SymmTile*)d_tl, ntiles, d_cnt);
            });
            CK(hipDeviceSynchronize());
            all = same(nm) && all;
            CK(hipMemset(rp[1], 0, (size_t)NV * rs * 8));
            CK(hipMemset(cp[1], 0, (size_t)NV * cs * 8));
        }
        // ---- loader / consumer waves (symm_lc_kernel.hpp)
        {
            int* d_err;
            CK(hipMalloc(&d_err, 4));
            CK(hipMemset(d_err, 0, 4));
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_symm_lc<SEG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SLC_LDS_BYTES));
            for (int wgs : {256, 128}) {
                CK(hipMemset(rp[1], 0, (size_t)NV * rs * 8));
                CK(hipMemset(cp[1], 0, (size_t)NV * cs * 8));
                snprintf(nm, sizeof nm, "k_symm_lc (2 loader + 6 MFMA waves), %d workgroups", wgs);
                timeit(nm, [&] {
                    CK(hipMemsetAsync(d_cnt, 0, 4, 0));
                    hipLaunchKernelGGL((k_symm_lc<SEG>), dim3((unsigned)wgs), dim3(512), SLC_LDS_BYTES, 0, (const double*)Q, ld, n, 0LL, n, (const double*)gT,
                                       16, rp[1], cp[1], rs, cs, (const DevState*)st, (const SymmTile*)d_tl, ntiles, d_cnt, d_err);
                });
                CK(hipDeviceSynchronize());
                int herr = 0;
                CK(hipMemcpy(&herr, d_err, 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(a.data(), cp[0], (size_t)NV * cs * 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(b.data(), cp[1], (size_t)NV * cs * 8, hipMemcpyDeviceToHost));
                const bool csame = memcmp(a.data(), b.data(), (size_t)NV * cs * 8) == 0;
                CK(hipMemcpy(a.data(), rp[0], (size_t)NV * rs * 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(b.data(), rp[1], (size_t)NV * rs * 8, hipMemcpyDeviceToHost));
                double mx = 0.0, df = 0.0;
                for (long long i = 0; i < (long long)NV * rs; ++i) {
                    mx = std::max(mx, std::fabs(a[i]));
                    df = std::max(df, std::fabs(a[i] - b[i]));
                }
                printf("   %s: hand-over failures %d; colpart %s; rowpart max |diff| / max |value| = %.3e %s\n", nm, herr,
                       csame ? "bit-identical" : "DIFFERENT", df / mx, (df / mx < 1e-13 && csame && !herr) ? "ok" : "WRONG");
                all = all && csame && !herr && df / mx < 1e-13;
            }
        }
        // ---- interleaved rounds in one process (the boxes and the order of the sections move these numbers by +-10 %)
        {
            int* d_err2;
            CK(hipMalloc(&d_err2, 4));
            CK(hipMemset(d_err2, 0, 4));
            const int NVAR = 9, ROUNDS = 25;
            std::vector<std::vector<float>> tms(NVAR);
            auto go = [&](int v) {
                switch (v) {
                case 0:
                    hipLaunchKernelGGL((k_symm_mfma<true, SEG>), dim3((unsigned)nstrips, (unsigned)nsegs), dim3(256), 0, 0, (const double*)Q, ld, n, 0LL, n,
                                       (const double*)gT, 16, rp[0], cp[0], rs, cs, (const DevState*)st);
                    break;
                case 1:
                case 2:
                    CK(hipMemsetAsync(d_cnt, 0, 4, 0));
                    hipLaunchKernelGGL((k_symm_q_reg<true, SEG>), dim3(v == 1 ? 512u : 768u), dim3(256), 0, 0, (const double*)Q, ld, n, 0LL, n,
                                       (const double*)gT, 16, rp[1], cp[1], rs, cs, (const DevState*)st, (const SymmTile*)d_tl, ntiles, d_cnt);
                    break;
                case 3:
                    CK(hipMemsetAsync(d_cnt, 0, 4, 0));
                    hipLaunchKernelGGL((k_symm_q_glds<SEG, 2>), dim3(512), dim3(256), lds2, 0, (const double*)Q, ld, n, 0LL, n, (const double*)gT, 16,
                                       rp[1], cp[1], rs, cs, (const DevState*)st, (const SymmTile*)d_tl, ntiles, d_cnt);
                    break;
                case 4:
                    CK(hipMemsetAsync(d_cnt, 0, 4, 0));
                    hipLaunchKernelGGL((k_symm_lc<SEG>), dim3(256), dim3(512), SLC_LDS_BYTES, 0, (const double*)Q, ld, n, 0LL, n, (const double*)gT, 16,
                                       rp[1], cp[1], rs, cs, (const DevState*)st, (const SymmTile*)d_tl, ntiles, d_cnt, d_err2);
                    break;
                case 5:
                    launch_glds<SEG, 2>((const double*)Q, 0, n, gT, 16, rp[1], cp[1], rs, cs, st);
                    break;
                case 6:
                case 7:
                case 8: {
                    const int budget = v == 6 ? 1 : (v == 7 ? 2 : 3);
                    CK(hipMemsetAsync(d_cnt, 0, 4, 0));
                    hipLaunchKernelGGL((k_symm_q_reg<true, SEG>), dim3((unsigned)((ntiles + budget - 1) / budget + 64)), dim3(256), 0, 0, (const double*)Q, ld, n, 0LL, n,
                                       (const double*)gT, 16, rp[1], cp[1], rs, cs, (const DevState*)st, (const SymmTile*)d_tl, ntiles, d_cnt, budget);
                    break;
                }
                }
            };
            for (int r = 0; r < ROUNDS + 2; ++r)
                for (int v = 0; v < NVAR; ++v) {
                    CK(hipEventRecord(e0, 0));
                    go(v);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float tm;
                    CK(hipEventElapsedTime(&tm, e0, e1));
                    if (r >= 2) tms[v].push_back(tm);
                }
            const char* names[NVAR] = {"k_symm_mfma (grid of tiles)", "k_symm_q_reg, 512 workgroups", "k_symm_q_reg, 768 workgroups",
                                       "k_symm_q_glds<D=2>, 512 workgroups", "k_symm_lc, 256 workgroups", "k_symm_glds<D=2> (grid of tiles)",
                                       "queue, ONE tile per workgroup (largest first)", "queue, at most 2 tiles per workgroup", "queue, at most 3 tiles per workgroup"};
            printf("interleaved, %d rounds, 16 gradients (the queue forms include the 4-byte memset of their counter):\n", ROUNDS);
            for (int v = 0; v < NVAR; ++v) {
                std::sort(tms[v].begin(), tms[v].end());
                printf("   %-40s median %.4f ms   min %.4f   p90 %.4f\n", names[v], tms[v][tms[v].size() / 2], tms[v][0], tms[v][tms[v].size() * 9 / 10]);
            }
        }
        const size_t lds3 = (size_t)4 * 3 * SGL_SLOT * sizeof(double);
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_symm_q_glds<SEG, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
        snprintf(nm, sizeof nm, "k_symm_q_glds<D=3>, %d tiles, 256 workgroups", ntiles);
        timeit(nm, [&] {
            CK(hipMemsetAsync(d_cnt, 0, 4, 0));
            hipLaunchKernelGGL((k_symm_q_glds<SEG, 3>), dim3(256), dim3(256), lds3, 0, (const double*)Q, ld, n, 0LL, n, (const double*)gT, 16, rp[1],
                               cp[1], rs, cs, (const DevState*)st, (const SymmTile*)d_tl, ntiles, d_cnt);
        });
        CK(hipDeviceSynchronize());
        all = same(nm) && all;
    }
    // the clock the chip holds in each form (MI355X_MICROARCH.md, DVFS give-back item 6): >= 2 s of back-to-back launches, then
    // one stamped launch; shader cycles per 100 MHz tick over the workgroups whose loop ran >= 20 us
    hipLaunchKernelGGL(k_pack_grads, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, 0, (const double*)g, n, 16, n, gT);
    auto clock_of = [&](const char* name, auto&& plain, auto&& stamped) {
        hipEvent_t a0, a1;
        CK(hipEventCreate(&a0));
        CK(hipEventCreate(&a1));
        float ms = 0;
        int reps = 0;
        CK(hipEventRecord(a0, 0));
        while (ms < 2000.0f) {
            for (int k = 0; k < 200; ++k) plain();
            reps += 200;
            CK(hipEventRecord(a1, 0));
            CK(hipEventSynchronize(a1));
            CK(hipEventElapsedTime(&ms, a0, a1));
        }
        stamped();
        CK(hipDeviceSynchronize());
        static unsigned long long h[8192][3];
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sgl_clk), sizeof h));
        std::vector<double> ghz;
        const long long nwg = std::min<long long>(8192, nstrips * nsegs);
        for (long long w = 0; w < nwg; ++w)
            if (h[w][1] >= 2000) ghz.push_back((double)h[w][0] / (double)h[w][1] * 0.1);
        std::sort(ghz.begin(), ghz.end());
        double cyc = 0, blks = 0;
        for (long long w = 0; w < nwg; ++w)
            if (h[w][2] >= 16) cyc += (double)h[w][0], blks += (double)h[w][2];
        printf("   [%s: %.0f shader cycles per block in wave 0's loop (workgroups with >= 16 blocks per wave)]\n", name, blks > 0 ? cyc / blks : 0.0);
        printf("%-34s %.4f ms per launch over %d back-to-back launches; in-kernel clock: median %.2f GHz (p10 %.2f, p90 %.2f, %zu workgroups)\n",
               name, ms / reps, reps, ghz.empty() ? 0.0 : ghz[ghz.size() / 2], ghz.empty() ? 0.0 : ghz[ghz.size() / 10],
               ghz.empty() ? 0.0 : ghz[ghz.size() * 9 / 10], ghz.size());
    };
    const double* Qc = Q;
#define CLKV(NAME, D, MODE, LV) \
    clock_of(NAME, [&] { launch_glds<SEG, D, MODE, false>(Qc, 0, n, gT, LV, rp[1], cp[1], rs, cs, st); }, \
             [&] { launch_glds<SEG, D, MODE, true>(Qc, 0, n, gT, LV, rp[1], cp[1], rs, cs, st); });
#define CLK(NAME, D, MODE) CLKV(NAME, D, MODE, 16)
    CLK("full kernel, D=2", 2, 0)
    CLK("loads only, D=2", 2, 1)
    CLK("MFMA only, D=2", 2, 2)
    CLK("loads never waited for, D=2", 2, 3)
    CLK("full kernel, D=3", 3, 0)
    CLK("4 loader + 4 MFMA waves, untied, D=2", 2, 6)
    CLK("4 loader + 4 MFMA waves, untied, D=3", 3, 6)
    CLK("2 loader waves on one SIMD + 6 MFMA waves, untied, D=2", 2, 7)
    CLK("2 loader waves on one SIMD + 6 MFMA waves, untied, D=3", 3, 7)
    CLK("MFMA only, D=3", 3, 2)
    CLKV("MFMA only, no stores, D=3", 3, 2, 0)
    CLK("MFMA only, no LDS reads, D=3", 3, 5)
    CLKV("MFMA only, no LDS reads, no stores, D=3", 3, 5, 0)
    CLKV("MFMA only, no LDS reads, no stores, D=2", 2, 5, 0)
#undef CLK
    printf(all ? "ALL IDENTICAL\n" : "MISMATCH\n");
    return all ? 0 : 1;
}
