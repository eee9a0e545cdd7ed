// symm_queue_check.hip -- GPU test (built with hipcc by tests/test_gpu_symm_queue.py): the matrix-core product pass in its queue forms
// (k_symm_mfma_q: tiles drawn from a counter, <= 16 gradients; k_symm_mfma_q2: two column tiles, <= 32) against the grid form round 3
// shipped (k_symm_mfma, one workgroup per tile), on a matrix whose upper triangle holds garbage (the kernels may read the lower
// triangle only), for whole matrices and for a row shard, full and ragged groups.  Partial sums compared BIT FOR BIT: per vector the
// arithmetic is the same, only who computes which tile (and when) differs.  Prints one JSON line per case.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../ellalgo-rs_amd/csrc/ell_kernels.hpp"
using namespace ellhip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void k_fill(double* p, long long m, unsigned long long salt) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x) {
        unsigned long long h = ((unsigned long long)i + salt) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29; h *= 0x94D049BB133111EBull; h ^= h >> 32;
        p[i] = (double)(h & 0xFFFFFFFFFFFFFull) / 4503599627370496.0 - 0.5;
    }
}

template <int SEG>
static bool run_case(long long n, long long row0, long long nrows, int lv, int wgs) {
    const long long ld = n + 16;
    const long long nstrips = (nrows + SYMV_H - 1) / SYMV_H, nsegs = (n + SEG - 1) / SEG, rs = nsegs * n, cs = nstrips * n;
    double *Q, *g, *gT16, *gT32, *rp[2], *cp[2];
    DevState* st;
    CK(hipMalloc(&Q, (size_t)nrows * ld * 8));
    CK(hipMalloc(&g, (size_t)32 * n * 8));
    CK(hipMalloc(&gT16, (size_t)2 * 16 * n * 8));
    CK(hipMalloc(&gT32, (size_t)32 * n * 8));
    for (int k = 0; k < 2; ++k) {
        CK(hipMalloc(&rp[k], (size_t)32 * rs * 8));
        CK(hipMalloc(&cp[k], (size_t)32 * cs * 8));
        CK(hipMemset(rp[k], 0, (size_t)32 * rs * 8));
        CK(hipMemset(cp[k], 0, (size_t)32 * cs * 8));
    }
    CK(hipMalloc(&st, sizeof(DevState)));
    CK(hipMemset(st, 0, sizeof(DevState)));
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, Q, nrows * ld, 1ull + (unsigned long long)n);
    hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, g, 32 * n, 77ull);
    std::vector<SymmTile> tl;
    for (long long I = nstrips - 1; I >= 0; --I)
        for (long long J = 0; J < nsegs; ++J)
            if (J * SEG <= row0 + I * SYMV_H + SYMV_H - 1) tl.push_back({(int)I, (int)J});
    auto blocks_of = [&](const SymmTile& t) {
        const long long r0 = row0 + (long long)t.I * SYMV_H, c0 = (long long)t.J * SEG;
        return (std::min<long long>(c0 + SEG, r0 + SYMV_H) - c0) / 16;
    };
    std::stable_sort(tl.begin(), tl.end(), [&](const SymmTile& a, const SymmTile& b) { return blocks_of(a) > blocks_of(b); });
    SymmTile* d_tl;
    unsigned* d_q;
    CK(hipMalloc(&d_tl, tl.size() * sizeof(SymmTile)));
    CK(hipMalloc(&d_q, 256));
    CK(hipMemcpy(d_tl, tl.data(), tl.size() * sizeof(SymmTile), hipMemcpyHostToDevice));
    const int ntiles = (int)tl.size();
    // reference: the grid form, 16 gradients at a time -> sets 0
    const int lva = std::min(lv, 16), lvb = lv - lva;
    hipLaunchKernelGGL(k_pack_grads, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, 0, (const double*)g, n, lva, n, gT16, (unsigned*)nullptr, 16);
    hipLaunchKernelGGL((k_symm_mfma<true, SEG>), dim3((unsigned)nstrips, (unsigned)nsegs), dim3(256), 0, 0, (const double*)Q, ld, n, row0, nrows,
                       (const double*)gT16, lva, rp[0], cp[0], rs, cs, (const DevState*)st);
    if (lvb > 0) {
        hipLaunchKernelGGL(k_pack_grads, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, 0, (const double*)(g + 16 * n), n, lvb, n, gT16 + 16 * n,
                           (unsigned*)nullptr, 16);
        hipLaunchKernelGGL((k_symm_mfma<true, SEG>), dim3((unsigned)nstrips, (unsigned)nsegs), dim3(256), 0, 0, (const double*)Q, ld, n, row0, nrows,
                           (const double*)(gT16 + 16 * n), lvb, rp[0] + 16 * rs, cp[0] + 16 * cs, rs, cs, (const DevState*)st);
    }
    // the queue forms -> sets 1 (k_pack_grads rewinds the counter, as in the product)
    if (lv <= 16) {
        hipLaunchKernelGGL(k_pack_grads, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, 0, (const double*)g, n, lv, n, gT16, d_q, 16);
        hipLaunchKernelGGL((k_symm_mfma_q<true, SEG>), dim3((unsigned)wgs), dim3(256), 0, 0, (const double*)Q, ld, n, row0, (const double*)gT16, lv, rp[1],
                           cp[1], rs, cs, (const DevState*)st, (const SymmTile*)d_tl, ntiles, d_q);
    } else {
        hipLaunchKernelGGL(k_pack_grads, dim3((unsigned)((n * 32 + 255) / 256)), dim3(256), 0, 0, (const double*)g, n, lv, n, gT32, d_q, 32);
        hipLaunchKernelGGL((k_symm_mfma_q2<true, SEG>), dim3((unsigned)wgs), dim3(256), 0, 0, (const double*)Q, ld, n, row0, (const double*)gT32, lv, rp[1],
                           cp[1], rs, cs, (const DevState*)st, (const SymmTile*)d_tl, ntiles, d_q);
    }
    CK(hipDeviceSynchronize());
    std::vector<double> a((size_t)32 * std::max(rs, cs)), b((size_t)32 * std::max(rs, cs));
    CK(hipMemcpy(a.data(), rp[0], (size_t)32 * rs * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), rp[1], (size_t)32 * rs * 8, hipMemcpyDeviceToHost));
    const bool rsame = memcmp(a.data(), b.data(), (size_t)32 * rs * 8) == 0;
    double rsum = 0.0;
    for (long long i = 0; i < 32 * rs; ++i) rsum += a[i] * a[i];
    CK(hipMemcpy(a.data(), cp[0], (size_t)32 * cs * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), cp[1], (size_t)32 * cs * 8, hipMemcpyDeviceToHost));
    const bool csame = memcmp(a.data(), b.data(), (size_t)32 * cs * 8) == 0;
    unsigned hq = 0;
    CK(hipMemcpy(&hq, d_q, 4, hipMemcpyDeviceToHost));
    printf("{\"n\": %lld, \"row0\": %lld, \"nrows\": %lld, \"seg\": %d, \"gradients\": %d, \"workgroups\": %d, \"tiles\": %d, \"rowpart_identical\": %s, "
           "\"colpart_identical\": %s, \"nonzero\": %s, \"queue_drawn\": %u}\n",
           n, row0, nrows, SEG, lv, wgs, ntiles, rsame ? "true" : "false", csame ? "true" : "false", rsum > 0.0 ? "true" : "false", hq);
    for (int k = 0; k < 2; ++k) {
        CK(hipFree(rp[k]));
        CK(hipFree(cp[k]));
    }
    CK(hipFree(Q)); CK(hipFree(g)); CK(hipFree(gT16)); CK(hipFree(gT32)); CK(hipFree(st)); CK(hipFree(d_tl)); CK(hipFree(d_q));
    return rsame && csame && rsum > 0.0;
}

int main() {
    bool ok = true;
    ok = run_case<2048>(4096, 0, 4096, 16, 512) && ok;
    ok = run_case<2048>(4096, 0, 4096, 5, 512) && ok;
    ok = run_case<2048>(4096, 0, 4096, 32, 512) && ok;
    ok = run_case<2048>(4096, 0, 4096, 20, 512) && ok;
    ok = run_case<2048>(4160, 0, 4160, 17, 64) && ok;      // fewer workgroups than tiles by far: every one draws many
    ok = run_case<2048>(8192, 2048, 4096, 32, 512) && ok;  // a symmetric row shard (rows 2048 .. 6143)
    ok = run_case<2048>(8192, 2048, 4096, 9, 300) && ok;
    ok = run_case<512>(2048, 0, 2048, 32, 512) && ok;      // the narrow segments of small shards
    ok = run_case<512>(2048, 512, 1024, 12, 512) && ok;
    return ok ? 0 : 1;
}
