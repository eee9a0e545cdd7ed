// tune_symvq.hip -- development tool: times the queue-driven lower-triangle GEMV tile phase (symvq_kernels.hpp) in
// several tile shapes / grid sizes against the static-grid k_symv of ell_kernels.hpp, checks every variant's y against
// the full-row GEMV, and prints each variant's per-workgroup timeline summary (first start, last end, spread of ends).
// Usage: tune_symvq [n] [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../../ellalgo-rs_amd/csrc/ell_kernels.hpp"
#include "symvq_kernels.hpp"

using namespace ellhip;

#define CK(x)                                                          \
    do {                                                               \
        hipError_t e = (x);                                            \
        if (e != hipSuccess) {                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));     \
            exit(1);                                                   \
        }                                                              \
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

// strip tiling: y[i] = sum_J rowpart[J][i] + sum_I colpart[I][i], one thread per column (check only)
__global__ void k_check_reduce(long long n, int H, int SEG, const double* rowpart, const double* colpart, double* y) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (long long J = 0; J <= i / SEG; ++J) s += rowpart[J * n + i];
    const long long nstrip = (n + H - 1) / H;
    for (long long I = i / H; I < nstrip; ++I) s += colpart[I * n + i];
    y[i] = s;
}
// run partition: y[i] = sum_{J <= i/512} rowpart[J][i] + sum_{pieces p of segment i/512} colpart[p][i mod 512]
__global__ void k_check_reduce_runs(long long n, const SymvqPlan* plan, const double* rowpart, const double* colpart, double* y) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int J = (int)(i / SQ_SEG);
    double s = 0.0;
    for (int j = 0; j <= J; ++j) s += rowpart[(long long)j * n + i];
    for (int p = plan->seg_pfirst[J]; p < plan->seg_pfirst[J + 1]; ++p) s += colpart[(long long)p * SQ_SEG + (i - (long long)J * SQ_SEG)];
    y[i] = s;
}

struct Variant {
    std::string name;
    std::function<void(hipStream_t)> launch;
    std::function<void(hipStream_t)> reduce;  // fills ychk
    unsigned G = 0;
    std::vector<float> ms;
};

int main(int argc, char** argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 16384;
    const int rounds = argc > 2 ? atoi(argv[2]) : 10;
    const long long ld = n + (argc > 3 ? atoll(argv[3]) : ((n % 512) == 0 ? 16 : 0));
    double *Q, *g, *yref, *ychk;
    DevState* st;
    CK(hipMalloc(&Q, (size_t)n * ld * 8));
    CK(hipMalloc(&g, n * 8));
    CK(hipMalloc(&yref, n * 8));
    CK(hipMalloc(&ychk, n * 8));
    CK(hipMalloc(&st, sizeof(DevState)));
    {
        std::vector<double> h((size_t)n);
        for (long long i = 0; i < n; ++i) h[i] = ((i * 2654435761u) % 1000) / 1000.0 - 0.5;
        CK(hipMemcpy(g, h.data(), n * 8, hipMemcpyHostToDevice));
        DevState s{};
        s.kappa = 1.0;
        CK(hipMemcpy(st, &s, sizeof s, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_fill_sym, dim3(4096), dim3(256), 0, 0, Q, ld, n);
        CK(hipDeviceSynchronize());
    }
    hipStream_t s;
    CK(hipStreamCreate(&s));
    // reference: full-row GEMV
    hipLaunchKernelGGL((k_sweep<4, 4, 2, true, false, true, false>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, Q, Q, ld, n, n,
                       0LL, (const double*)nullptr, g, yref, st, 0);
    CK(hipStreamSynchronize(s));
    std::vector<double> href((size_t)n), hchk((size_t)n);
    CK(hipMemcpy(href.data(), yref, n * 8, hipMemcpyDeviceToHost));

    double *rowpart, *colpart;
    CK(hipMalloc(&rowpart, (size_t)((n + 127) / 128) * n * 8));   // up to SEG = 128
    CK(hipMalloc(&colpart, (size_t)((n + 15) / 16) * n * 8));     // down to H = 16 / many small pieces
    unsigned long long* stamps;
    CK(hipMalloc(&stamps, 4 * 4096 * 8));
    SymvqCtl* ctl;
    CK(hipMalloc(&ctl, sizeof(SymvqCtl)));
    CK(hipMemset(ctl, 0, sizeof(SymvqCtl)));

    std::vector<Variant> vs;
    {
        Variant v;
        v.name = "k_symv static 64x2048 RW2 (round 1)";
        v.launch = [=](hipStream_t q) {
            dim3 grid((unsigned)((n + SYMV_H - 1) / SYMV_H), (unsigned)((n + SYMV_SEG - 1) / SYMV_SEG));
            hipLaunchKernelGGL((k_symv<2, true, 0, SYMV_SEG>), grid, dim3(256), 0, q, Q, ld, n, 0LL, n, g, rowpart, colpart, st);
        };
        v.reduce = [=](hipStream_t q) {
            hipLaunchKernelGGL(k_check_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, q, n, (int)SYMV_H, (int)SYMV_SEG, rowpart, colpart, ychk);
        };
        vs.push_back(v);
    }
    SymvqPlan hplan;
    std::vector<SymvqPiece> hpieces;
    symvq_make_plan(n, hplan, hpieces, argc > 4 ? atof(argv[4]) : 0.15, argc > 5 ? atoll(argv[5]) : 64);
    {
        long long mn = 1LL << 60, mx = 0;
        for (int w = 0; w < SQ_RUNS; ++w) {
            long long c = 0;
            for (int p = hplan.run_first[w]; p < hplan.run_first[w + 1]; ++p)
                for (int r = hpieces[p].ra; r < hpieces[p].rb; ++r) {
                    const long long dg = r - (long long)hpieces[p].J * SQ_SEG;
                    c += (dg < SQ_SEG - 1 ? dg : SQ_SEG - 1) / 128 + 1;
                }
            mn = std::min(mn, c);
            mx = std::max(mx, c);
        }
        printf("plan: %d segments, %d pieces (%d dynamic), run cost %lld..%lld chunk loads\n", hplan.nseg, hplan.npiece, hplan.ndyn, mn, mx);
    }
    SymvqPlan* dplan;
    SymvqPiece* dpieces;
    CK(hipMalloc(&dplan, sizeof(SymvqPlan)));
    CK(hipMalloc(&dpieces, hpieces.size() * sizeof(SymvqPiece)));
    CK(hipMemcpy(dplan, &hplan, sizeof hplan, hipMemcpyHostToDevice));
    CK(hipMemcpy(dpieces, hpieces.data(), hpieces.size() * sizeof(SymvqPiece), hipMemcpyHostToDevice));
#define RV(RW, GG, WPS)                                                                                            \
    {                                                                                                              \
        Variant v;                                                                                                 \
        v.G = GG;                                                                                                  \
        v.name = "symvq runs RW" #RW " G" #GG " w" #WPS;                                                           \
        v.launch = [=](hipStream_t q) {                                                                            \
            hipLaunchKernelGGL((k_symvq_runs<RW, true, WPS>), dim3(GG), dim3(256), 0, q, Q, ld, n, g, rowpart,     \
                               colpart, dplan, dpieces, ctl, stamps);                                                   \
        };                                                                                                         \
        v.reduce = [=](hipStream_t q) {                                                                            \
            hipLaunchKernelGGL(k_check_reduce_runs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, q, n, dplan,  \
                               rowpart, colpart, ychk);                                                            \
        };                                                                                                         \
        vs.push_back(v);                                                                                           \
    }
    RV(2, 1024, 4) RV(4, 768, 3) RV(1, 1024, 5)

    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    printf("n=%lld ld=%lld rounds=%d   4n^2 = %.1f MB\n", n, ld, rounds, 4.0 * n * n / 1e6);
    // correctness of every variant first
    for (auto& v : vs) {
        CK(hipMemsetAsync(rowpart, 0xff, (size_t)((n + 127) / 128) * n * 8, s));  // NaN-poison: every slot read must have been written
        CK(hipMemsetAsync(colpart, 0xff, (size_t)((n + 15) / 16) * n * 8, s));
        v.launch(s);
        v.reduce(s);
        CK(hipStreamSynchronize(s));
        CK(hipGetLastError());
        CK(hipMemcpy(hchk.data(), ychk, n * 8, hipMemcpyDeviceToHost));
        double err = 0.0, sc = 0.0;
        for (long long i = 0; i < n; ++i) {
            const double d = std::fabs(hchk[i] - href[i]);
            err = (d > err || d != d) ? (d != d ? INFINITY : d) : err;
            sc = std::max(sc, std::fabs(href[i]));
        }
        printf("check %-46s max|y - y_gemv| / max|y| = %.3e %s\n", v.name.c_str(), err / sc, err / sc < 1e-12 ? "ok" : "MISMATCH");
    }
    std::vector<unsigned long long> hst(4 * 4096);
    for (int r = 0; r < rounds + 1; ++r) {
        for (auto& v : vs) {
            // realistic cache state: a different pass over Q precedes every timed launch
            hipLaunchKernelGGL((k_sweep<4, 4, 2, true, false, true, false>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, Q, Q, ld,
                               n, n, 0LL, (const double*)nullptr, g, yref, st, 1);
            CK(hipEventRecord(a, s));
            v.launch(s);
            CK(hipEventRecord(b, s));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            if (r > 0) v.ms.push_back(ms);
            if (r == rounds && v.G) {  // timeline of the last round
                CK(hipMemcpy(hst.data(), stamps, 4 * v.G * 8, hipMemcpyDeviceToHost));
                unsigned long long t0 = ~0ull, t1 = 0, b_max = 0, dmin = ~0ull, dmax = 0;
                std::vector<double> ends, sends;
                for (unsigned w = 0; w < v.G; ++w) {
                    t0 = std::min(t0, hst[4 * w]);
                    b_max = std::max(b_max, hst[4 * w]);
                    t1 = std::max(t1, hst[4 * w + 2]);
                    dmin = std::min(dmin, hst[4 * w + 3]);
                    dmax = std::max(dmax, hst[4 * w + 3]);
                }
                for (unsigned w = 0; w < v.G; ++w) {
                    ends.push_back((hst[4 * w + 2] - t0) / 100.0);
                    sends.push_back((hst[4 * w + 1] - t0) / 100.0);
                }
                std::sort(ends.begin(), ends.end());
                std::sort(sends.begin(), sends.end());
                auto pct = [](const std::vector<double>& a, double f) { return a[(size_t)(f * (a.size() - 1))]; };
                // wall_clock64 ticks at 100 MHz
                printf("   timeline %-26s static part ends: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f | all ends: min %.1f p50 %.1f p90 %.1f max %.1f us | dyn pieces per wg %llu..%llu\n",
                       v.name.c_str(), pct(sends, 0), pct(sends, .1), pct(sends, .5), pct(sends, .9), pct(sends, 1), pct(ends, 0),
                       pct(ends, .5), pct(ends, .9), pct(ends, 1), dmin, dmax);
            }
        }
    }
    CK(hipGetLastError());
    const double bytes = 4.0 * (double)n * (double)n;
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const double med = v.ms[v.ms.size() / 2], mn = v.ms.front(), mx = v.ms.back();
        printf("%-46s med %.4f ms  min %.4f  max %.4f  %7.1f GB/s (med) %7.1f GB/s (best)\n", v.name.c_str(), med, mn, mx,
               bytes / med / 1e6, bytes / mn / 1e6);
    }
    return 0;
}
