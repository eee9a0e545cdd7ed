// tune_ell.hip -- in-process A/B timing of the Ell sweep-kernel variants (development tool, not shipped).
// Usage: tune_ell [n] [rounds] [pad]   Prints median/min ms and algorithmic GB/s per variant; variants
// are timed interleaved (round-robin) in one process (cdna_hip_programming.md 5.4 rule 24).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../ellalgo-rs_amd/csrc/ell_kernels.hpp"

using namespace ellhip;

#define CK(x)                                                          \
    do {                                                               \
        hipError_t e = (x);                                            \
        if (e != hipSuccess) {                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));     \
            exit(1);                                                   \
        }                                                              \
    } while (0)

struct Variant {
    std::string name;
    double bytes;
    std::function<void(hipStream_t, int)> launch;
    std::vector<float> ms;
};

int main(int argc, char** argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 16384;
    const int rounds = argc > 2 ? atoi(argv[2]) : 10;
    const long long pad = argc > 3 ? atoll(argv[3]) : 0;
    const long long ld = n + pad;
    double *Q, *Q2, *g, *gt, *gt2;
    DevState* st;
    CK(hipMalloc(&Q, (size_t)n * ld * 8));
    CK(hipMalloc(&Q2, (size_t)n * ld * 8));
    CK(hipMemset(Q2, 0, (size_t)n * ld * 8));
    CK(hipMalloc(&g, n * 8));
    CK(hipMalloc(&gt, n * 8));
    CK(hipMalloc(&gt2, n * 8));
    CK(hipMalloc(&st, sizeof(DevState)));
    {
        std::vector<double> h((size_t)n);
        for (long long i = 0; i < n; ++i) h[i] = 1e-3 * ((i * 2654435761u) % 1000) / 1000.0;
        CK(hipMemcpy(g, h.data(), n * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(gt, h.data(), n * 8, hipMemcpyHostToDevice));
        DevState s{};
        s.kappa = 1.0;
        s.ratio = 1e-9;
        s.scale = 1.0;
        s.apply = 1;
        CK(hipMemcpy(st, &s, sizeof s, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_fill_diag, dim3(2048), dim3(256), 0, 0, Q, ld, n, n, 0LL, (const double*)nullptr);
        CK(hipDeviceSynchronize());
    }
    hipStream_t s;
    CK(hipStreamCreate(&s));
    std::vector<Variant> vs;
    const double n2 = (double)n * (double)n;
#define SWEEP(RW, UNR, NT, R1, GV, OUT)                                                                     \
    vs.push_back({std::string(R1 && GV ? "fused" : (R1 ? "rank1" : "gemv ")) + " RW" #RW " UNR" #UNR +      \
                      (NT ? " nt" : "   ") + (OUT ? " out" : ""),                                           \
                  (R1 ? 16.0 : 8.0) * n2, [=](hipStream_t q, int rev) {                                      \
                      unsigned grid = (unsigned)((n + RW - 1) / RW);                                        \
                      hipLaunchKernelGGL((k_sweep<RW, UNR, 2, NT, R1, GV, false>), dim3(grid), dim3(256), 0, q, Q, \
                                         OUT ? Q2 : Q, ld, n, n, 0LL, gt, g, gt2, st, rev);                  \
                  }, {}});
    SWEEP(1, 4, true, false, true, false) SWEEP(2, 4, true, false, true, false) SWEEP(4, 2, true, false, true, false)
    SWEEP(4, 4, true, false, true, false) SWEEP(8, 2, true, false, true, false) SWEEP(8, 1, true, false, true, false)
    SWEEP(1, 4, false, false, true, false) SWEEP(4, 4, false, false, true, false)
    SWEEP(1, 4, true, true, false, false) SWEEP(1, 8, true, true, false, false) SWEEP(2, 4, true, true, false, false)
    SWEEP(2, 8, true, true, false, false) SWEEP(4, 2, true, true, false, false) SWEEP(4, 4, true, true, false, false)
    SWEEP(8, 2, true, true, false, false) SWEEP(1, 8, false, true, false, false) SWEEP(4, 4, false, true, false, false)
    SWEEP(2, 8, true, true, false, true) SWEEP(2, 4, true, true, false, true)
    SWEEP(1, 4, true, true, true, false) SWEEP(1, 8, true, true, true, false) SWEEP(2, 4, true, true, true, false)
    SWEEP(2, 8, true, true, true, false) SWEEP(4, 2, true, true, true, false) SWEEP(4, 4, true, true, true, false)
    SWEEP(8, 2, true, true, true, false) SWEEP(1, 8, false, true, true, false) SWEEP(4, 4, false, true, true, false)
    SWEEP(2, 8, true, true, true, true) SWEEP(2, 4, true, true, true, true)

    double *rowpart, *colpart;
    CK(hipMalloc(&rowpart, (size_t)((n + SYMV_SEG - 1) / SYMV_SEG) * n * 8));
    CK(hipMalloc(&colpart, (size_t)((n + SYMV_H - 1) / SYMV_H) * n * 8));
#define SYMVV(RW, NT, ABL)                                                                                   \
    vs.push_back({std::string("symv  RW" #RW) + (NT ? " nt" : "   ") + " abl" #ABL, 4.0 * n2, [=](hipStream_t q, int) { \
                      dim3 grid((unsigned)((n + SYMV_H - 1) / SYMV_H), (unsigned)((n + SYMV_SEG - 1) / SYMV_SEG));       \
                      hipLaunchKernelGGL((k_symv<RW, NT, ABL>), grid, dim3(256), 0, q, Q, ld, n, 0LL, n, g, rowpart, colpart, st); \
                  }, {}});
    SYMVV(2, true, 0) SYMVV(2, true, 1) SYMVV(2, true, 2) SYMVV(4, true, 0) SYMVV(4, true, 1) SYMVV(2, false, 0)
    vs.push_back({"symv reduce", 0.0, [=](hipStream_t q, int) {
                      hipLaunchKernelGGL(k_symv_reduce<0>, dim3((unsigned)((n + 127) / 128)), dim3(256), 0, q, n, 0LL, n, (long long)SYMV_SEG, rowpart, colpart, gt2, st, (const double*)nullptr, (const double*)nullptr, (double*)nullptr);
                  }, {}});
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    // Each timed launch follows a launch of a full sweep in the opposite direction (as in a real update
    // sequence), so the cache state is realistic.
    for (int r = 0; r < rounds + 1; ++r) {
        for (auto& v : vs) {
            hipLaunchKernelGGL((k_sweep<2, 4, 2, true, true, true, false>), dim3((unsigned)((n + 1) / 2)), dim3(256), 0, s,
                               Q, Q, ld, n, n, 0LL, gt, g, gt2, st, 0);
            CK(hipEventRecord(a, s));
            v.launch(s, 1);
            CK(hipEventRecord(b, s));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            if (r > 0) v.ms.push_back(ms);
        }
    }
    CK(hipGetLastError());
    printf("n=%lld ld=%lld rounds=%d\n", n, ld, rounds);
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const double med = v.ms[v.ms.size() / 2], mn = v.ms.front();
        printf("%-24s med %.4f ms  min %.4f ms  %7.1f GB/s (med) %7.1f GB/s (best)\n", v.name.c_str(), med, mn,
               v.bytes / med / 1e6, v.bytes / mn / 1e6);
    }
    return 0;
}
