// tune_ell.hip -- in-process A/B timing of the Ell kernel variants (development tool, not shipped).
// Usage: tune_ell [n] [rounds]   Prints avg/min ms and algorithmic GB/s per variant; variants are
// timed interleaved (round-robin) in one process (cdna_hip_programming.md 5.4 rule 24).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../ellalgo-rs_amd/csrc/ell_kernels.hpp"

using namespace ellhip;

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

// calibration kernels: plain linear streams over the same buffers
template <bool NT>
__global__ __launch_bounds__(256) void k_copy_linear(const double2_t* __restrict__ a, double2_t* __restrict__ b, long long nvec) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        double2_t v0, v1, v2, v3;
        if (NT) { v0 = __builtin_nontemporal_load(a + i); v1 = __builtin_nontemporal_load(a + i + stride); v2 = __builtin_nontemporal_load(a + i + 2 * stride); v3 = __builtin_nontemporal_load(a + i + 3 * stride); }
        else { v0 = a[i]; v1 = a[i + stride]; v2 = a[i + 2 * stride]; v3 = a[i + 3 * stride]; }
        if (NT) { __builtin_nontemporal_store(v0, b + i); __builtin_nontemporal_store(v1, b + i + stride); __builtin_nontemporal_store(v2, b + i + 2 * stride); __builtin_nontemporal_store(v3, b + i + 3 * stride); }
        else { b[i] = v0; b[i + stride] = v1; b[i + 2 * stride] = v2; b[i + 3 * stride] = v3; }
    }
    for (; i < nvec; i += stride) b[i] = a[i];
}
template <bool NT>
__global__ __launch_bounds__(256) void k_read_linear(const double2_t* __restrict__ a, double* __restrict__ out, long long nvec) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    double s = 0;
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        double2_t v0, v1, v2, v3;
        if (NT) { v0 = __builtin_nontemporal_load(a + i); v1 = __builtin_nontemporal_load(a + i + stride); v2 = __builtin_nontemporal_load(a + i + 2 * stride); v3 = __builtin_nontemporal_load(a + i + 3 * stride); }
        else { v0 = a[i]; v1 = a[i + stride]; v2 = a[i + 2 * stride]; v3 = a[i + 3 * stride]; }
        s += v0.x + v0.y + v1.x + v1.y + v2.x + v2.y + v3.x + v3.y;
    }
    if (s == 123.456) out[0] = s;
}

struct Variant {
    std::string name;
    int kind;  // 0 gemv, 1 rank1
    std::function<void(hipStream_t)> launch;
    std::vector<float> ms;
};

int main(int argc, char** argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 16384;
    const int rounds = argc > 2 ? atoi(argv[2]) : 10;
    const long long pad = argc > 3 ? atoll(argv[3]) : 0;
    const long long ld = n + pad;
    double *Q, *Q2, *g, *gt;
    DevState* st;
    CK(hipMalloc(&Q, (size_t)n * ld * 8));
    CK(hipMalloc(&Q2, (size_t)n * ld * 8));
    CK(hipMemset(Q2, 0, (size_t)n * ld * 8));
    CK(hipMalloc(&g, n * 8));
    CK(hipMalloc(&gt, n * 8));
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
#define GEMV(RW, UNR, NT)                                                                              \
    vs.push_back({std::string("gemv  RW" #RW " UNR" #UNR) + (NT ? " nt" : "   "), 0, [=](hipStream_t q) {              \
                      unsigned grid = (unsigned)((n + 4 * RW - 1) / (4 * RW));                         \
                      hipLaunchKernelGGL((k_gemv<RW, UNR, 2, NT>), dim3(grid), dim3(256), 0, q, Q, ld, n, n, g, gt, st); \
                  }, {}});
#define RANK1X(RW, UNR, NT, REV, OUT)                                                                        \
    vs.push_back({std::string("rank1 RW" #RW " UNR" #UNR) + (NT ? " nt" : "   ") + (REV ? " rev" : " fwd") + (OUT ? " out" : " inp"), 1, [=](hipStream_t q) { \
                      unsigned grid = (unsigned)((n + 4 * RW - 1) / (4 * RW));                         \
                      hipLaunchKernelGGL((k_rank1<RW, UNR, 2, false, NT, REV>), dim3(grid), dim3(256), 0, q, Q, OUT ? Q2 : Q, ld, n, n, 0LL, gt, st); \
                  }, {}});
#define RANK1(RW, UNR, NT, REV) RANK1X(RW, UNR, NT, REV, false)
    GEMV(1, 4, false) GEMV(1, 8, false) GEMV(2, 2, false) GEMV(2, 4, false) GEMV(2, 8, false)
    GEMV(4, 1, false) GEMV(4, 2, false) GEMV(4, 4, false) GEMV(8, 1, false) GEMV(8, 2, false)
    GEMV(2, 4, true) GEMV(4, 2, true) GEMV(4, 4, true) GEMV(8, 2, true)
    RANK1(1, 4, false, true) RANK1(1, 8, false, true) RANK1(2, 2, false, true) RANK1(2, 4, false, true)
    RANK1(4, 1, false, true) RANK1(4, 2, false, true) RANK1(4, 4, false, true) RANK1(8, 1, false, true) RANK1(8, 2, false, true)
    RANK1(2, 4, true, true) RANK1(4, 2, true, true) RANK1(4, 4, true, true) RANK1(8, 2, true, true)
    RANK1(4, 2, false, false) RANK1(4, 2, true, false)
    RANK1X(4, 2, false, true, true) RANK1X(4, 2, true, true, true) RANK1X(2, 4, true, true, true) RANK1X(4, 4, true, true, true) RANK1X(1, 4, false, true, true)

    const long long nvec = n * ld / 2;
    for (int blocks : {2048, 4096, 8192, 16384}) {
        vs.push_back({"copy linear    " + std::to_string(blocks), 1, [=](hipStream_t q) { hipLaunchKernelGGL(k_copy_linear<false>, dim3(blocks), dim3(256), 0, q, (const double2_t*)Q, (double2_t*)Q2, nvec); }, {}});
        vs.push_back({"copy linear nt " + std::to_string(blocks), 1, [=](hipStream_t q) { hipLaunchKernelGGL(k_copy_linear<true>, dim3(blocks), dim3(256), 0, q, (const double2_t*)Q, (double2_t*)Q2, nvec); }, {}});
        vs.push_back({"copy inplace nt " + std::to_string(blocks), 1, [=](hipStream_t q) { hipLaunchKernelGGL(k_copy_linear<true>, dim3(blocks), dim3(256), 0, q, (const double2_t*)Q, (double2_t*)Q, nvec); }, {}});
        vs.push_back({"read linear    " + std::to_string(blocks), 0, [=](hipStream_t q) { hipLaunchKernelGGL(k_read_linear<false>, dim3(blocks), dim3(256), 0, q, (const double2_t*)Q, gt, nvec); }, {}});
        vs.push_back({"read linear nt " + std::to_string(blocks), 0, [=](hipStream_t q) { hipLaunchKernelGGL(k_read_linear<true>, dim3(blocks), dim3(256), 0, q, (const double2_t*)Q, gt, nvec); }, {}});
    }
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    // Each timed launch is preceded by a launch of the OTHER pass (as in a real update sequence), so
    // cache state is realistic: gemv before rank1, rank1 before gemv.
    auto other = [&](int kind, hipStream_t q) {
        if (kind == 0) hipLaunchKernelGGL((k_rank1<4, 2, 2, false, false, true>), dim3((unsigned)((n + 15) / 16)), dim3(256), 0, q, Q, Q, ld, n, n, 0LL, gt, st);
        else hipLaunchKernelGGL((k_gemv<4, 2, 2, false>), dim3((unsigned)((n + 15) / 16)), dim3(256), 0, q, Q, ld, n, n, g, gt, st);
    };
    for (int r = 0; r < rounds + 1; ++r) {
        for (auto& v : vs) {
            other(v.kind, s);
            CK(hipEventRecord(a, s));
            v.launch(s);
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
        const double bytes = (v.kind == 0 ? 8.0 : 16.0) * n * n;
        printf("%-28s med %.4f ms  min %.4f ms  %7.1f GB/s (med) %7.1f GB/s (best)\n", v.name.c_str(), med, mn,
               bytes / med / 1e6, bytes / mn / 1e6);
    }
    return 0;
}
