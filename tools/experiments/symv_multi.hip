// symv_multi.hip -- round-3 experiment: the lower-triangle GEMV for LV queued gradients in one pass over Q
// (k_symv_multi, ell_kernels.hpp) against LV launches of k_symv: time per launch / per vector, and every vector's
// partial sums compared bit for bit with what k_symv writes for that gradient alone.
// Usage: symv_multi [n] [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../ellalgo-rs_amd/csrc/ell_kernels.hpp"
#include "symm_split_kernel.hpp"
#include "symm32_kernel.hpp"

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
        const unsigned long long lo = r < c ? r : c, hi = r < c ? c : r;
        unsigned long long h = (hi * 0x9E3779B97F4A7C15ull) ^ (lo * 0xBF58476D1CE4E5B9ull);
        h ^= h >> 29; h *= 0x94D049BB133111EBull; h ^= h >> 32;
        Q[i] = (double)(h & 0xFFFFF) / 1048576.0 - 0.5 + (r == c ? 2.0 : 0.0);
    }
}
__global__ void k_fill_vec(double* g, long long m) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x) {
        unsigned long long h = (unsigned long long)(i + 12345) * 0xD6E8FEB86659FD93ull;
        h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32;
        g[i] = (double)(h & 0xFFFFF) / 1048576.0 - 0.5;
    }
}

__global__ void k_scale_vec(double* g, long long m) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x)
        g[i] = g[i] * 0.7390851332151607 + 1e-3 / (double)(i % 977 + 3);
}

constexpr int MAXL = 8;

template <int RW, int SEG, int LV>
static void launch_multi(const double* Q, long long ld, long long n, const double* g, double* rowpart, double* colpart,
                         long long rs, long long cs, const DevState* st) {
    const unsigned nstrips = (unsigned)((n + SYMV_H - 1) / SYMV_H), nsegs = (unsigned)((n + SEG - 1) / SEG);
    hipLaunchKernelGGL((k_symv_multi<RW, true, SEG, LV>), dim3(nstrips, nsegs), dim3(256), 0, 0, Q, ld, n, 0LL, n, g, n,
                       rowpart, colpart, rs, cs, st);
}

static long long n, ld;
static int rounds;
static double *Q, *g, *rp, *cp, *rp1, *cp1;
static DevState* st;
static hipEvent_t e0, e1;

template <typename F>
static void timeit(const char* name, int lv, F&& fn) {
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
    printf("%-40s med %.4f ms  min %.4f  per vector %.4f ms   %.1f GB/s of 4n^2 per launch\n", name, med, ms[0], med / lv,
           4.0 * n * n / 1e9 / (med * 1e-3));
    CK(hipGetLastError());
}

template <int SEG>
static void run_seg() {
    const long long nstrips = (n + SYMV_H - 1) / SYMV_H, nsegs = (n + SEG - 1) / SEG;
    const long long rs = nsegs * n, cs = nstrips * n;
    printf("-- tiles 64 x %d: partial sums per vector %.1f MB\n", SEG, (rs + cs) * 8.0 / 1e6);
    auto single = [&](int l) {
        hipLaunchKernelGGL((k_symv<2, true, 0, SEG>), dim3((unsigned)nstrips, (unsigned)nsegs), dim3(256), 0, 0,
                           (const double*)Q, ld, n, 0LL, n, (const double*)(g + l * n), rp1 + l * rs, cp1 + l * cs,
                           (const DevState*)st);
    };
    CK(hipMemset(rp1, 0, (size_t)MAXL * rs * 8));
    CK(hipMemset(cp1, 0, (size_t)MAXL * cs * 8));
    for (int l = 0; l < MAXL; ++l) single(l);
    CK(hipDeviceSynchronize());
    std::vector<double> a((size_t)std::max(rs, cs)), b((size_t)std::max(rs, cs));
    auto check = [&](const char* name, int lv) {
        bool ok = true;
        for (int l = 0; l < lv && ok; ++l) {
            CK(hipMemcpy(a.data(), rp + l * rs, (size_t)rs * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), rp1 + l * rs, (size_t)rs * 8, hipMemcpyDeviceToHost));
            ok = ok && memcmp(a.data(), b.data(), (size_t)rs * 8) == 0;
            CK(hipMemcpy(a.data(), cp + l * cs, (size_t)cs * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), cp1 + l * cs, (size_t)cs * 8, hipMemcpyDeviceToHost));
            ok = ok && memcmp(a.data(), b.data(), (size_t)cs * 8) == 0;
        }
        printf("   check %-31s partial sums of %d vectors vs k_symv<2, nt, 0, %d>: %s\n", name, lv, SEG,
               ok ? "bit-identical" : "DIFFERENT");
    };
    char nm[64];
    snprintf(nm, sizeof nm, "k_symv<2, nt, 0, %d> x1", SEG);
    timeit(nm, 1, [&] { single(0); });
#define RUN(RW, LV)                                                                                  \
    {                                                                                                \
        CK(hipMemset(rp, 0, (size_t)MAXL * rs * 8));                                                 \
        CK(hipMemset(cp, 0, (size_t)MAXL * cs * 8));                                                 \
        snprintf(nm, sizeof nm, "k_symv_multi<RW=%d, SEG=%d, LV=%d>", RW, SEG, LV);                  \
        timeit(nm, LV, [&] { launch_multi<RW, SEG, LV>(Q, ld, n, g, rp, cp, rs, cs, st); });         \
        CK(hipDeviceSynchronize());                                                                  \
        check(nm, LV);                                                                               \
    }
    RUN(2, 1)
    RUN(2, 2)
    RUN(2, 3)
    RUN(2, 4)
    if (SEG <= 1024) {
        RUN(4, 4)
        RUN(2, 6)
        RUN(2, 8)
        RUN(4, 8)
    }
#undef RUN
}

int main(int argc, char** argv) {
    n = argc > 1 ? atoll(argv[1]) : 16384;
    rounds = argc > 2 ? atoi(argv[2]) : 8;
    ld = n + 16;
    const long long nstrips = (n + SYMV_H - 1) / SYMV_H;
    const long long rsmax = ((n + 511) / 512) * n, cs = nstrips * n;
    CK(hipMalloc(&Q, (size_t)n * ld * 8));
    CK(hipMalloc(&g, (size_t)MAXL * n * 8));
    CK(hipMalloc(&rp, (size_t)MAXL * rsmax * 8));
    CK(hipMalloc(&cp, (size_t)MAXL * cs * 8));
    CK(hipMalloc(&rp1, (size_t)MAXL * rsmax * 8));
    CK(hipMalloc(&cp1, (size_t)MAXL * cs * 8));
    CK(hipMalloc(&st, sizeof(DevState)));
    CK(hipMemset(st, 0, sizeof(DevState)));
    hipLaunchKernelGGL(k_fill_sym, dim3(4096), dim3(256), 0, 0, Q, ld, n);
    hipLaunchKernelGGL(k_fill_vec, dim3(256), dim3(256), 0, 0, g, (long long)MAXL * n);
    CK(hipDeviceSynchronize());
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("n=%lld ld=%lld rounds=%d   4n^2 = %.1f MB\n", n, ld, rounds, 4.0 * n * n / 1e6);
    run_seg<2048>();
    if (argc > 3) {
        run_seg<1024>();
        run_seg<512>();
    }
    // ---- the matrix-core kernel: up to 16 gradients per pass (reference partial sums: k_symv<2, nt, 0, 2048>, tolerance)
    {
        constexpr int SEG = 2048;
        const long long nsegs = (n + SEG - 1) / SEG;
        const long long rs = nsegs * n;
        const int NV = 16;
        double *g16, *gT, *rpm, *cpm;
        CK(hipMalloc(&g16, (size_t)NV * n * 8));
        CK(hipMalloc(&gT, (size_t)NV * n * 8));
        CK(hipMalloc(&rpm, (size_t)NV * rs * 8));
        CK(hipMalloc(&cpm, (size_t)NV * cs * 8));
        hipLaunchKernelGGL(k_fill_vec, dim3(256), dim3(256), 0, 0, g16, (long long)NV * n);
        hipLaunchKernelGGL(k_scale_vec, dim3(256), dim3(256), 0, 0, g16, (long long)NV * n);  // (full 53-bit mantissas: sums round)
        CK(hipMemset(rpm, 0, (size_t)NV * rs * 8));
        CK(hipMemset(cpm, 0, (size_t)NV * cs * 8));
        std::vector<double> a((size_t)std::max(rs, cs)), b((size_t)std::max(rs, cs));
        for (int lv : {16, 12, 5}) {
            char nm[64];
            snprintf(nm, sizeof nm, "k_pack_grads + k_symm_mfma, %d vectors", lv);
            timeit(nm, lv, [&] {
                hipLaunchKernelGGL(k_pack_grads, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, 0, (const double*)g16, n, lv, n, gT);
                hipLaunchKernelGGL((k_symm_mfma<true, SEG>), dim3((unsigned)nstrips, (unsigned)nsegs), dim3(256), 0, 0,
                                   (const double*)Q, ld, n, 0LL, n, (const double*)gT, lv, rpm, cpm, rs, cs, (const DevState*)st);
            });
            snprintf(nm, sizeof nm, "k_symm_mfma_split alone, %d vectors", lv);
            timeit(nm, lv, [&] {
                hipLaunchKernelGGL((k_symm_mfma_split<true, SEG>), dim3((unsigned)nstrips, (unsigned)nsegs), dim3(256), 0, 0,
                                   (const double*)Q, ld, n, (const double*)gT, lv, rpm, cpm, rs, cs, (const DevState*)st);
            });
            CK(hipDeviceSynchronize());
            {
                double worst = 0.0;
                for (int l = 0; l < lv; l += (lv > 4 ? 5 : 1)) {
                    CK(hipMemset(rp1, 0, (size_t)rs * 8));
                    CK(hipMemset(cp1, 0, (size_t)cs * 8));
                    hipLaunchKernelGGL((k_symv<2, true, 0, SEG>), dim3((unsigned)nstrips, (unsigned)nsegs), dim3(256), 0, 0,
                                       (const double*)Q, ld, n, 0LL, n, (const double*)(g16 + l * n), rp1, cp1, (const DevState*)st);
                    CK(hipDeviceSynchronize());
                    for (int which = 0; which < 2; ++which) {
                        const long long m = which ? cs : rs;
                        CK(hipMemcpy(a.data(), (which ? cpm + l * cs : rpm + l * rs), (size_t)m * 8, hipMemcpyDeviceToHost));
                        CK(hipMemcpy(b.data(), which ? cp1 : rp1, (size_t)m * 8, hipMemcpyDeviceToHost));
                        double mx = 0.0, df = 0.0;
                        for (long long i = 0; i < m; ++i) {
                            mx = std::max(mx, std::fabs(b[i]));
                            df = std::max(df, std::fabs(a[i] - b[i]));
                        }
                        worst = std::max(worst, df / mx);
                    }
                }
                printf("   check (split): max |diff| / max |value| = %.3e %s\n", worst, worst < 1e-13 ? "ok" : "WRONG");
            }
            CK(hipMemset(rpm, 0, (size_t)NV * rs * 8));
            CK(hipMemset(cpm, 0, (size_t)NV * cs * 8));
            snprintf(nm, sizeof nm, "k_symm_mfma alone, %d vectors", lv);
            timeit(nm, lv, [&] {
                hipLaunchKernelGGL((k_symm_mfma<true, SEG>), dim3((unsigned)nstrips, (unsigned)nsegs), dim3(256), 0, 0,
                                   (const double*)Q, ld, n, 0LL, n, (const double*)gT, lv, rpm, cpm, rs, cs, (const DevState*)st);
            });
            CK(hipDeviceSynchronize());
            double worst = 0.0;
            for (int l = 0; l < lv; ++l) {
                CK(hipMemset(rp1, 0, (size_t)rs * 8));
                CK(hipMemset(cp1, 0, (size_t)cs * 8));
                hipLaunchKernelGGL((k_symv<2, true, 0, SEG>), dim3((unsigned)nstrips, (unsigned)nsegs), dim3(256), 0, 0,
                                   (const double*)Q, ld, n, 0LL, n, (const double*)(g16 + l * n), rp1, cp1, (const DevState*)st);
                CK(hipDeviceSynchronize());
                for (int which = 0; which < 2; ++which) {
                    const long long m = which ? cs : rs;
                    CK(hipMemcpy(a.data(), (which ? cpm + l * cs : rpm + l * rs), (size_t)m * 8, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(b.data(), which ? cp1 : rp1, (size_t)m * 8, hipMemcpyDeviceToHost));
                    double mx = 0.0, df = 0.0;
                    for (long long i = 0; i < m; ++i) {
                        mx = std::max(mx, std::fabs(b[i]));
                        df = std::max(df, std::fabs(a[i] - b[i]));
                    }
                    worst = std::max(worst, df / mx);
                }
            }
            printf("   check: partial sums of %d vectors vs k_symv, max |diff| / max |value| = %.3e %s\n", lv, worst,
                   worst < 1e-13 ? "ok" : "WRONG");
        }
    }
    // ---- 32 gradients per pass (two N-tiles)
    {
        constexpr int SEG = 2048;
        const long long nsegs = (n + SEG - 1) / SEG;
        const long long rs = nsegs * n;
        const int NV = 32;
        double *g32, *gT, *rpm, *cpm;
        CK(hipMalloc(&g32, (size_t)NV * n * 8));
        CK(hipMalloc(&gT, (size_t)NV * n * 8));
        CK(hipMalloc(&rpm, (size_t)NV * rs * 8));
        CK(hipMalloc(&cpm, (size_t)NV * cs * 8));
        hipLaunchKernelGGL(k_fill_vec, dim3(256), dim3(256), 0, 0, g32, (long long)NV * n);
        hipLaunchKernelGGL(k_scale_vec, dim3(256), dim3(256), 0, 0, g32, (long long)NV * n);
        CK(hipMemset(rpm, 0, (size_t)NV * rs * 8));
        CK(hipMemset(cpm, 0, (size_t)NV * cs * 8));
        std::vector<double> a((size_t)std::max(rs, cs)), b((size_t)std::max(rs, cs));
        for (int lv : {32, 20}) {
            char nm[64];
            hipLaunchKernelGGL(k_pack_grads32, dim3((unsigned)((n * 32 + 255) / 256)), dim3(256), 0, 0, (const double*)g32, n, lv, n, gT);
            snprintf(nm, sizeof nm, "k_symm_mfma32 alone, %d vectors", lv);
            timeit(nm, lv, [&] {
                hipLaunchKernelGGL((k_symm_mfma32<true, SEG>), dim3((unsigned)nstrips, (unsigned)nsegs), dim3(256), 0, 0,
                                   (const double*)Q, ld, n, (const double*)gT, lv, rpm, cpm, rs, cs, (const DevState*)st);
            });
            CK(hipDeviceSynchronize());
            double worst = 0.0;
            for (int l : {0, 7, 16, lv - 1}) {
                CK(hipMemset(rp1, 0, (size_t)rs * 8));
                CK(hipMemset(cp1, 0, (size_t)cs * 8));
                hipLaunchKernelGGL((k_symv<2, true, 0, SEG>), dim3((unsigned)nstrips, (unsigned)nsegs), dim3(256), 0, 0,
                                   (const double*)Q, ld, n, 0LL, n, (const double*)(g32 + l * n), rp1, cp1, (const DevState*)st);
                CK(hipDeviceSynchronize());
                for (int which = 0; which < 2; ++which) {
                    const long long m = which ? cs : rs;
                    CK(hipMemcpy(a.data(), (which ? cpm + l * cs : rpm + l * rs), (size_t)m * 8, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(b.data(), which ? cp1 : rp1, (size_t)m * 8, hipMemcpyDeviceToHost));
                    double mx = 0.0, df = 0.0;
                    for (long long i = 0; i < m; ++i) {
                        mx = std::max(mx, std::fabs(b[i]));
                        df = std::max(df, std::fabs(a[i] - b[i]));
                    }
                    worst = std::max(worst, df / mx);
                }
            }
            printf("   check (32): max |diff| / max |value| = %.3e %s\n", worst, worst < 1e-13 ? "ok" : "WRONG");
        }
    }
    return 0;
}
