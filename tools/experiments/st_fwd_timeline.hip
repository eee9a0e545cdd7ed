// st_fwd_timeline.hip -- where does a step of the persistent EllStable forward solve (k_st_fwd_persist) go?
// Wall-clock stamps (100 MHz) per workgroup and step, through the ST_STAMP hooks of ellstable_kernels.hpp:
//   0 step start | 4 (last step) own block fetched and parked | 1 the row block's w has arrived | 2 products + column
//   sums done | 3 products written back | 7 (wave 2) own w handed to the stores | 5 diagonal block done | 6 all done
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int TL_MAXB = 128;
__device__ unsigned long long g_stamps[2 * TL_MAXB][TL_MAXB + 1][16];
#define ST_STAMP(kb, slot) do { if (threadIdx.x == 0) g_stamps[blockIdx.x][(kb)][(slot)] = wall_clock64(); } while (0)
#define ST_STAMP_LANE0(kb, slot) do { if ((threadIdx.x & 63) == 0) g_stamps[blockIdx.x][(kb)][(slot)] = wall_clock64(); } while (0)
#include "../../ellalgo-rs_amd/csrc/ellstable_kernels.hpp"
using namespace ellhip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 16384;
    const long long ld = n + 16, nb = (n + SB - 1) / SB;
    if (nb > TL_MAXB) return 1;
    // factor: small strict upper entries, diagonal in [0.5, 1.5); the scratch triangle may hold anything
    std::vector<double> hM((size_t)n * ld);
    unsigned long long x = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (double)(x >> 11) / 9007199254740992.0; };
    for (long long i = 0; i < n; ++i)
        for (long long j = 0; j < n; ++j) hM[i * ld + j] = (i == j) ? 0.5 + rnd() : (rnd() - 0.5) * 0.2 / 128.0;
    std::vector<double> hg(n);
    for (auto& v : hg) v = rnd() - 0.5;
    double *M, *g, *w, *z, *gg; int *flags, *err; DevState* st;
    CK(hipMalloc(&M, hM.size() * 8)); CK(hipMemcpy(M, hM.data(), hM.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&g, n * 8)); CK(hipMemcpy(g, hg.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&w, n * 8)); CK(hipMalloc(&z, n * 8)); CK(hipMalloc(&gg, n * 8));
    CK(hipMalloc(&flags, 2 * nb * 4)); CK(hipMemset(flags, 0, 2 * nb * 4));
    CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
    CK(hipMalloc(&st, sizeof(DevState))); CK(hipMemset(st, 0, sizeof(DevState)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 1; rep <= 4; ++rep) {
        hipLaunchKernelGGL(k_st_arm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, w, n);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_st_fwd_persist, dim3((unsigned)nb), dim3(256), 0, 0, M, ld, n, g, w, z, gg, flags, err, rep, st);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    static unsigned long long h[2 * TL_MAXB][TL_MAXB + 1][16];
    CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof(h)));
    printf("n = %lld: k_st_fwd_persist %.1f us, %.2f us per block; err %d\n", n, ms * 1000, ms * 1000 / nb, herr);
    const unsigned long long t0 = h[0][0][5];
    auto us = [&](unsigned long long t) { return ((double)t - (double)t0) / 100.0; };
    // chain: when block s's values were handed to the stores (slot 7), and the last step of workgroup s against it
    std::vector<double> T, late, seen, comp, diag, parkt, wb, steplen, flagdelay;
    for (long long s = 2; s < nb; ++s) {
        const double pub_prev = us(h[s - 1][s - 1][7]), pub = us(h[s][s][7]);
        const auto& L = h[s][s - 1];
        T.push_back(pub - pub_prev);
        late.push_back(us(L[4]) - pub_prev);          // > 0: the workgroup was still fetching / parking when the values came out
        seen.push_back(us(L[1]) - std::max(pub_prev, us(L[4])));  // poll delay once both sides are ready
        comp.push_back(us(L[2]) - us(L[1]));
        diag.push_back(pub - us(L[2]));
        parkt.push_back(us(L[4]) - us(L[0]));         // rows requested + own block fetched + parked
        if (s >= 3) {
            const auto& P = h[s][s - 2];              // the step before the last one
            wb.push_back(us(P[3]) - us(P[2]));
            steplen.push_back(us(P[3]) - us(P[0]));
            flagdelay.push_back(us(P[1]) - std::max(us(h[s - 2][s - 2][7]), us(P[0])));
        }
    }
    auto stat = [](const char* name, std::vector<double> a) {
        std::sort(a.begin(), a.end());
        double sum = 0; for (double v : a) sum += v;
        printf("  %-58s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f us\n", name, sum / a.size(), a[a.size() / 10], a[a.size() / 2], a[a.size() * 9 / 10]);
    };
    stat("chain period (hand-over s-1 -> hand-over s)", T);
    stat("last step: parked minus previous hand-over (>0 = late)", late);
    stat("last step: values seen after both sides ready", seen);
    stat("last step: products + column sums", comp);
    stat("last step: barrier + diagonal block until hand-over", diag);
    stat("last step: rows requested, own block fetched and parked", parkt);
    stat("step before: flag seen after its hand-over / step start", flagdelay);
    stat("step before: products written back", wb);
    stat("step before: whole step", steplen);
    // one workgroup in the middle, step by step
    const long long s = nb / 2;
    printf("  workgroup %lld, its last 4 steps (us since block 0 was solved): \n", s);
    for (long long kb = s - 4; kb < s; ++kb) {
        if (kb < 0) continue;
        printf("    kb %3lld: start %.2f", kb, us(h[s][kb][0]));
        if (kb == s - 1) printf(" parked %.2f", us(h[s][kb][4]));
        printf(" values %.2f (block handed over at %.2f) sums %.2f end %.2f\n", us(h[s][kb][1]), us(h[kb][kb][7]), us(h[s][kb][2]), us(h[s][kb][3]));
    }
    printf("    own: handed over %.2f, diagonal block done %.2f, all done %.2f\n", us(h[s][s][7]), us(h[s][s][5]), us(h[s][s][6]));
    // ---- the same solve with helper workgroups (k_st_fwd_helped): chain workgroup 2s+1, helper 2s
    double* hpart; CK(hipMalloc(&hpart, n * 8));
    for (int rep = 5; rep <= 8; ++rep) {
        hipLaunchKernelGGL(k_st_arm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, w, n);
        hipLaunchKernelGGL(k_st_arm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, hpart, n);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_st_fwd_helped, dim3((unsigned)(2 * nb)), dim3(256), 0, 0, M, ld, n, g, w, hpart, z, gg, flags, err, rep, st);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof(h)));
    printf("n = %lld: k_st_fwd_helped %.1f us, %.2f us per block; err %d\n", n, ms * 1000, ms * 1000 / nb, herr);
    {
        const unsigned long long t1 = h[1][0][7];
        auto uh = [&](unsigned long long t) { return ((double)t - (double)t1) / 100.0; };
        std::vector<double> T, hp_wait, w_wait, sums, diag, ready_before, h_flag, h_vals, h_apply, h_wb, h_pub_after;
        for (long long s = 3; s < nb; ++s) {
            const auto& C = h[2 * s + 1][s];
            const double pub_prev = uh(h[2 * s - 1][s - 1][7]), pub = uh(C[7]);
            T.push_back(pub - pub_prev);
            ready_before.push_back(pub_prev - uh(C[0]));     // > 0: parked and waiting before the previous hand-over
            hp_wait.push_back(uh(C[1]) - pub_prev);          // helper's sums in hand, relative to the previous hand-over
            w_wait.push_back(uh(C[2]) - std::max(pub_prev, uh(C[1])));
            sums.push_back(uh(C[3]) - uh(C[2]));
            diag.push_back(pub - uh(C[3]));
            const auto& H = h[2 * s][s - 2];                 // the helper's last step (row block s - 2)
            const double pub_pp = uh(h[2 * s - 3][s - 2][7]);
            h_flag.push_back(uh(H[1]) - std::max(pub_pp, uh(H[0])));
            h_vals.push_back(uh(H[2]) - uh(H[1]));
            h_apply.push_back(uh(H[3]) - uh(H[2]));
            h_wb.push_back(uh(H[4]) - uh(H[3]));
            h_pub_after.push_back(uh(H[3]) - pub_pp);        // helper's sums stored, after block s-2's hand-over
        }
        stat("chain period", T);
        stat("chain wg: ready this long before the previous hand-over", ready_before);
        stat("chain wg: helper's sums seen, after the previous hand-over", hp_wait);
        stat("chain wg: w values seen after (hand-over, sums)", w_wait);
        stat("chain wg: products + column sums", sums);
        stat("chain wg: diagonal block until hand-over", diag);
        stat("helper last step: flag seen after hand-over s-2 / step start", h_flag);
        stat("helper last step: published w loaded", h_vals);
        stat("helper last step: rows requested + applied + sums stored", h_apply);
        stat("helper last step: write-back", h_wb);
        stat("helper: sums stored this long after hand-over s-2", h_pub_after);
        std::vector<double> d_ld, d_a, d_b1, d_mini, d_b2, d_b, d_pub;
        for (long long s = 3; s < nb; ++s) {
            const auto& C = h[2 * s + 1][s];
            d_ld.push_back(uh(C[8]) - uh(C[3]));    // wave 0: its 64 columns from LDS
            d_a.push_back(uh(C[9]) - uh(C[8]));     // chain A
            d_b1.push_back(uh(C[10]) - uh(C[9]));   // barrier, seen by wave 1
            d_mini.push_back(uh(C[11]) - uh(C[10]));
            d_b2.push_back(uh(C[12]) - uh(C[11]));  // barrier, seen by wave 2
            d_b.push_back(uh(C[13]) - uh(C[12]));   // chain B
            d_pub.push_back(uh(C[7]) - uh(C[13]));  // 128 write-through stores issued
        }
        stat("  diagonal block: columns from LDS (wave 0)", d_ld);
        stat("  diagonal block: chain A", d_a);
        stat("  diagonal block: barrier 1", d_b1);
        stat("  diagonal block: mini panel (wave 1)", d_mini);
        stat("  diagonal block: barrier 2", d_b2);
        stat("  diagonal block: chain B (wave 2)", d_b);
        stat("  diagonal block: publish stores issued", d_pub);
        std::vector<double> tail;
        for (long long s = 8; s < nb; ++s) tail.push_back(uh(h[2 * s + 1][s][6]) - uh(h[2 * s + 1][s][5]));
        stat("chain wg: products written after the own hand-over", tail);
    }
    return 0;
}
