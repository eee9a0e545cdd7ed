// copy_calib.hip -- what does a plain 16-byte-per-lane copy reach on this box, next to the read-modify-write passes of
// the engine?  (Round-1 verdict, item 9: the guide quotes 6.29 TB/s for a float4 copy; the rank-1 / apply passes run at
// 5.2-5.4 TB/s.)  Every variant moves the same 2 x 2 GiB (read + write) as the rank-1 pass at n = 16384.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../../ellalgo-rs_amd/csrc/ell_kernels.hpp"
using namespace ellhip;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float float4_t __attribute__((ext_vector_type(4)));

template <bool NTL, bool NTS, int UNR>
__global__ __launch_bounds__(256) void k_copy(const float4_t* __restrict__ in, float4_t* __restrict__ out, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256 * UNR;
    for (size_t i = (size_t)blockIdx.x * 256 * UNR + threadIdx.x; i < n16; i += stride) {
        float4_t v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (i + 256 * u < n16) v[u] = NTL ? __builtin_nontemporal_load(in + i + 256 * u) : in[i + 256 * u];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (i + 256 * u < n16) { if (NTS) __builtin_nontemporal_store(v[u], out + i + 256 * u); else out[i + 256 * u] = v[u]; }
    }
}
// each workgroup copies ONE contiguous slab (the guide's "one block per chunk" shape)
template <bool NTL, int UNR>
__global__ __launch_bounds__(256) void k_copy_slab(const float4_t* __restrict__ in, float4_t* __restrict__ out, size_t n16) {
    const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
    const size_t lo = per * blockIdx.x, hi = std::min(n16, lo + per);
    for (size_t i = lo + threadIdx.x; i < hi; i += 256 * UNR) {
        float4_t v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) if (i + 256 * u < hi) v[u] = NTL ? __builtin_nontemporal_load(in + i + 256 * u) : in[i + 256 * u];
#pragma unroll
        for (int u = 0; u < UNR; ++u) if (i + 256 * u < hi) out[i + 256 * u] = v[u];
    }
}
template <bool NTL, int UNR>
__global__ __launch_bounds__(256) void k_scale_inplace(double2_t* __restrict__ x, size_t n16, double a) {
    const size_t stride = (size_t)gridDim.x * 256 * UNR;
    for (size_t i = (size_t)blockIdx.x * 256 * UNR + threadIdx.x; i < n16; i += stride) {
        double2_t v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) if (i + 256 * u < n16) v[u] = NTL ? __builtin_nontemporal_load(x + i + 256 * u) : x[i + 256 * u];
#pragma unroll
        for (int u = 0; u < UNR; ++u) if (i + 256 * u < n16) { v[u].x = v[u].x - a * v[u].y; x[i + 256 * u] = v[u]; }
    }
}

int main(int argc, char** argv) {
    const long long n = 16384, ld = n + 16;
    const int rounds = argc > 1 ? atoi(argv[1]) : 10;
    const size_t bytes = (size_t)n * ld * 8, n16 = bytes / 16;
    double *a, *b, *g, *gt;
    DevState* st;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&g, n * 8)); CK(hipMalloc(&gt, n * 8)); CK(hipMalloc(&st, sizeof(DevState)));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(g, 0, n * 8)); CK(hipMemset(gt, 0, n * 8));
    DevState hs{}; hs.kappa = 1.0; hs.ratio = 1e-9; hs.scale = 1.0; hs.apply = 1;
    CK(hipMemcpy(st, &hs, sizeof hs, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    struct V { std::string name; double bytes; std::function<void()> go; std::vector<float> ms; };
    std::vector<V> vs;
    const float4_t* in = (const float4_t*)a; float4_t* out = (float4_t*)b;
#define COPY(NTL, NTS, UNR, GRID) vs.push_back({std::string("copy float4 ") + (NTL ? "nt-load " : "load ") + (NTS ? "nt-store " : "store ") + "x" #UNR " grid " #GRID, 2.0 * bytes, [=]() { hipLaunchKernelGGL((k_copy<NTL, NTS, UNR>), dim3(GRID), dim3(256), 0, s, in, out, n16); }, {}});
    COPY(false, false, 1, 2048) COPY(false, false, 4, 2048) COPY(true, false, 4, 2048) COPY(true, true, 4, 2048) COPY(false, true, 4, 2048)
    COPY(false, false, 4, 8192) COPY(true, false, 4, 8192) COPY(true, false, 8, 4096) COPY(true, false, 4, 65536) COPY(true, false, 2, 1024)
#define SLAB(NTL, UNR, GRID) vs.push_back({std::string("copy slabs ") + (NTL ? "nt-load " : "load ") + "x" #UNR " grid " #GRID, 2.0 * bytes, [=]() { hipLaunchKernelGGL((k_copy_slab<NTL, UNR>), dim3(GRID), dim3(256), 0, s, in, out, n16); }, {}});
    SLAB(false, 4, 2048) SLAB(true, 4, 2048) SLAB(true, 4, 8192)
#define INPL(NTL, UNR, GRID) vs.push_back({std::string("in-place x -= a*y ") + (NTL ? "nt-load " : "load ") + "x" #UNR " grid " #GRID, 2.0 * bytes, [=]() { hipLaunchKernelGGL((k_scale_inplace<NTL, UNR>), dim3(GRID), dim3(256), 0, s, (double2_t*)a, n16, 1e-30); }, {}});
    INPL(false, 4, 2048) INPL(true, 4, 2048) INPL(true, 4, 8192)
    vs.push_back({"engine rank-1 pass k_sweep<2,8,nt,R1>", 16.0 * n * n, [=]() { hipLaunchKernelGGL((k_sweep<2, 8, 2, true, true, false, false>), dim3((unsigned)(n / 2)), dim3(256), 0, s, (const double*)a, a, ld, n, n, 0LL, (const double*)gt, (const double*)g, gt, (const DevState*)st, 0); }, {}});
    vs.push_back({"engine GEMV pass k_sweep<4,4,nt,GV> (read only)", 8.0 * n * n, [=]() { hipLaunchKernelGGL((k_sweep<4, 4, 2, true, false, true, false>), dim3((unsigned)(n / 4)), dim3(256), 0, s, (const double*)a, a, ld, n, n, 0LL, (const double*)nullptr, (const double*)g, gt, (const DevState*)st, 0); }, {}});
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < rounds + 1; ++r)
        for (auto& v : vs) {
            CK(hipEventRecord(e0, s)); v.go(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) v.ms.push_back(ms);
        }
    CK(hipGetLastError());
    printf("2 GiB buffers (n = 16384, ld = n + 16), %d rounds, variants interleaved; GB/s = (read + written bytes) / time\n", rounds);
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const double med = v.ms[v.ms.size() / 2], mn = v.ms.front();
        printf("%-52s med %.4f ms  min %.4f ms  %7.1f GB/s (med) %7.1f GB/s (best)\n", v.name.c_str(), med, mn, v.bytes / med / 1e6, v.bytes / mn / 1e6);
    }
    return 0;
}
