// symm32_queue.hip -- round-4 experiment: 32 gradients per matrix-core pass from the tile queue (k_symm_q32) against the grid form
// (k_symm_mfma32) and against two 16-wide passes of the product's k_symm_mfma_q; interleaved rounds, partial sums compared.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../ellalgo-rs_amd/csrc/ell_kernels.hpp"
#include "symm32_kernel.hpp"
using namespace ellhip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_fill(double* Q, long long m, unsigned long long salt) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x) {
        unsigned long long h = ((unsigned long long)i + salt) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29; h *= 0x94D049BB133111EBull; h ^= h >> 32;
        Q[i] = (double)(h & 0xFFFFFFFFFFFFFull) / 4503599627370496.0 - 0.5;
    }
}

int main(int argc, char** argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 16384, ld = n + 16;
    constexpr int SEG = 2048;
    const long long nstrips = n / SYMV_H, nsegs = (n + SEG - 1) / SEG, rs = nsegs * n, cs = nstrips * n;
    double *Q, *g, *gT16, *gT32, *rp[2], *cp[2];
    DevState* st;
    CK(hipMalloc(&Q, (size_t)n * ld * 8));
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
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, Q, n * ld, 1ull);
    hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, g, 32 * n, 77ull);
    std::vector<SymmTile> tl;
    for (int I = (int)nstrips - 1; I >= 0; --I)
        for (int J = 0; J < (int)nsegs; ++J)
            if ((long long)J * SEG <= (long long)I * SYMV_H + SYMV_H - 1) tl.push_back({I, J});
    auto blocks_of = [&](const SymmTile& t) {
        const long long r0 = (long long)t.I * SYMV_H, c0 = (long long)t.J * SEG;
        return (std::min<long long>(c0 + SEG, r0 + SYMV_H) - c0) / 16;
    };
    std::stable_sort(tl.begin(), tl.end(), [&](const SymmTile& a, const SymmTile& b) { return blocks_of(a) > blocks_of(b); });
    SymmTile* d_tl;
    unsigned* d_q;
    CK(hipMalloc(&d_tl, tl.size() * sizeof(SymmTile)));
    CK(hipMalloc(&d_q, 256));
    CK(hipMemcpy(d_tl, tl.data(), tl.size() * sizeof(SymmTile), hipMemcpyHostToDevice));
    const int ntiles = (int)tl.size();
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int lv : {32, 20}) {
        hipLaunchKernelGGL(k_pack_grads32, dim3((unsigned)((n * 32 + 255) / 256)), dim3(256), 0, 0, (const double*)g, n, lv, n, gT32);
        const int lva = std::min(lv, 16), lvb = lv - lva;
        hipLaunchKernelGGL(k_pack_grads, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, 0, (const double*)g, n, lva, n, gT16, (unsigned*)nullptr);
        hipLaunchKernelGGL(k_pack_grads, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, 0, (const double*)(g + 16 * n), n, lvb, n, gT16 + 16 * n,
                           (unsigned*)nullptr);
        auto go = [&](int v) {
            switch (v) {
            case 0:  // two 16-wide passes of the product's queue kernel -> sets 0 (the reference)
                CK(hipMemsetAsync(d_q, 0, 256, 0));
                hipLaunchKernelGGL((k_symm_mfma_q<true, SEG>), dim3(768), dim3(256), 0, 0, (const double*)Q, ld, n, 0LL, (const double*)gT16, lva, rp[0],
                                   cp[0], rs, cs, (const DevState*)st, (const SymmTile*)d_tl, ntiles, d_q);
                hipLaunchKernelGGL((k_symm_mfma_q<true, SEG>), dim3(768), dim3(256), 0, 0, (const double*)Q, ld, n, 0LL, (const double*)(gT16 + 16 * n), lvb,
                                   rp[0] + 16 * rs, cp[0] + 16 * cs, rs, cs, (const DevState*)st, (const SymmTile*)d_tl, ntiles, d_q + 32);
                break;
            case 1:
                hipLaunchKernelGGL((k_symm_mfma32<true, SEG>), dim3((unsigned)nstrips, (unsigned)nsegs), dim3(256), 0, 0, (const double*)Q, ld, n,
                                   (const double*)gT32, lv, rp[1], cp[1], rs, cs, (const DevState*)st);
                break;
            case 4:
                CK(hipMemsetAsync(d_q, 0, 256, 0));
                hipLaunchKernelGGL((k_symm_q32<true, SEG, 1>), dim3(512u), dim3(256), 0, 0, (const double*)Q, ld, n, (const double*)gT32, lv,
                                   rp[1], cp[1], rs, cs, (const DevState*)st, (const SymmTile*)d_tl, ntiles, d_q);
                break;
            default:
                CK(hipMemsetAsync(d_q, 0, 256, 0));
                hipLaunchKernelGGL((k_symm_q32<true, SEG>), dim3(v == 2 ? 512u : 768u), dim3(256), 0, 0, (const double*)Q, ld, n, (const double*)gT32, lv,
                                   rp[1], cp[1], rs, cs, (const DevState*)st, (const SymmTile*)d_tl, ntiles, d_q);
                break;
            }
        };
        const char* names[5] = {"2 x k_symm_mfma_q (16 + rest), 768 wgs", "k_symm_mfma32 (grid of tiles)", "k_symm_q32, 512 workgroups", "k_symm_q32, 768 workgroups", "k_symm_q32, no occupancy hint, 512 wgs"};
        std::vector<std::vector<float>> tms(5);
        for (int r = 0; r < 22; ++r)
            for (int v = 0; v < 5; ++v) {
                CK(hipEventRecord(e0, 0));
                go(v);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float tm;
                CK(hipEventElapsedTime(&tm, e0, e1));
                if (r >= 2) tms[v].push_back(tm);
            }
        printf("n = %lld, %d gradients, interleaved rounds:\n", n, lv);
        for (int v = 0; v < 5; ++v) {
            std::sort(tms[v].begin(), tms[v].end());
            printf("   %-42s median %.4f ms   min %.4f\n", names[v], tms[v][tms[v].size() / 2], tms[v][0]);
        }
        // the 32-wide partial sums against the two 16-wide passes
        go(0);
        CK(hipMemset(rp[1], 0, (size_t)32 * rs * 8));
        CK(hipMemset(cp[1], 0, (size_t)32 * cs * 8));
        go(3);
        CK(hipDeviceSynchronize());
        std::vector<double> a((size_t)std::max(rs, cs)), b((size_t)std::max(rs, cs));
        double worst = 0.0;
        bool csame = true;
        for (int l = 0; l < lv; ++l) {
            CK(hipMemcpy(a.data(), cp[0] + l * cs, (size_t)cs * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), cp[1] + l * cs, (size_t)cs * 8, hipMemcpyDeviceToHost));
            csame = csame && memcmp(a.data(), b.data(), (size_t)cs * 8) == 0;
            CK(hipMemcpy(a.data(), rp[0] + l * rs, (size_t)rs * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), rp[1] + l * rs, (size_t)rs * 8, hipMemcpyDeviceToHost));
            if (memcmp(a.data(), b.data(), (size_t)rs * 8) != 0) {
                double mx = 0, df = 0;
                for (long long i = 0; i < rs; ++i) mx = std::max(mx, std::fabs(a[i])), df = std::max(df, std::fabs(a[i] - b[i]));
                worst = std::max(worst, df / mx);
            }
        }
        printf("   k_symm_q32 vs the 16-wide passes: colpart %s, rowpart max |diff| / max |value| %.3e\n", csame ? "bit-identical" : "DIFFERENT", worst);
    }
    return 0;
}
