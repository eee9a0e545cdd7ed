// resident_bench.hip -- milestone 1 of the on-die persistent update (VERDICT r02 "next" item 5): does a launch that parks
// the lower triangle of Q (n = 4096: 64 MiB) in the register files of all CUs and runs K updates with two grid-wide
// hand-offs each stay under 20 us per update?  Runs csrc/resident_kernels.hpp's k_ell_resident on a synthetic queue of deep
// cuts from Q0 = I, checks the state after K0 cuts against a plain host loop (src/ell.rs:97-137 in double, row-major) and
// times K cuts per launch.
// Usage: resident_bench [n] [K] [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../ellalgo-rs_amd/csrc/resident_kernels.hpp"

using namespace ellhip;

#define CK(x)                                                      \
    do {                                                           \
        hipError_t e = (x);                                        \
        if (e != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); \
            exit(1);                                               \
        }                                                          \
    } while (0)

int main(int argc, char** argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 4096;
    const int K = argc > 2 ? atoi(argv[2]) : 200;
    const int rounds = argc > 3 ? atoi(argv[3]) : 5;
    const int K0 = std::min(K, 12);  // cuts checked against the host loop
    const long long ld = n;
    const int T = (int)((n + RS_TS - 1) / RS_TS);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    int R = 0, S = 0;
    for (int r = 1; r <= RS_RMAX && !R; ++r) {
        const int sr = (T + r - 1) / r;
        if (sr <= RS_SMAX && sr * (sr + 1) / 2 <= cus) { R = r; S = sr; }
    }
    if (argc > 4) { R = atoi(argv[4]); S = (T + R - 1) / R; }
    if (!R || S > RS_SMAX || S * (S + 1) / 2 > cus) {
        printf("n=%lld does not fit the register files\n", n);
        return 1;
    }
    const int G = S * (S + 1) / 2, NV = 2 * R * RS_TS;
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, R == 1 ? (const void*)k_ell_resident<1> : (R == 2 ? (const void*)k_ell_resident<2> : (const void*)k_ell_resident<3>), RS_THREADS, 0));
    printf("n=%lld T=%d R=%d S=%d grid=%d CUs=%d occupancy=%d\n", n, T, R, S, G, cus, occ);
    if (occ < 1) return 1;
    // queue: unit gradients, deep cuts beta ~ U[0, 0.05)
    std::vector<double> grads((size_t)K * n), xc0((size_t)n, 0.0);
    std::vector<CutParams> cps((size_t)K);
    unsigned long long s = 0x9E3779B97F4A7C15ull;
    auto u = [&]() {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        return (double)(s >> 11) / 9007199254740992.0;
    };
    for (int k = 0; k < K; ++k) {
        double nrm = 0.0;
        for (long long i = 0; i < n; ++i) {
            const double x = u() - 0.5;
            grads[(size_t)k * n + i] = x;
            nrm += x * x;
        }
        nrm = std::sqrt(nrm);
        for (long long i = 0; i < n; ++i) grads[(size_t)k * n + i] /= nrm;
        cps[k] = CutParams{0, 0, 0.05 * u(), 0.0};
    }
    double *dQ, *dg, *dxc, *dpart, *dy, *dom, *dtsq;
    CutParams* dcp;
    int* dstat;
    DevState* dst;
    unsigned* dctr;
    CK(hipMalloc(&dQ, (size_t)n * ld * 8));
    CK(hipMalloc(&dg, (size_t)K * n * 8));
    CK(hipMalloc(&dxc, n * 8));
    CK(hipMalloc(&dpart, (size_t)2 * G * NV * 8));
    CK(hipMalloc(&dom, (size_t)2 * G * 8));
    dy = nullptr;
    CK(hipMalloc(&dtsq, K * 8));
    CK(hipMalloc(&dcp, K * sizeof(CutParams)));
    CK(hipMalloc(&dstat, K * sizeof(int)));
    CK(hipMalloc(&dst, sizeof(DevState)));
    CK(hipMalloc(&dctr, RS_BAR_WORDS * sizeof(unsigned)));
    unsigned long long* dstamps = nullptr;
#ifdef RS_TIMELINE
    CK(hipMalloc(&dstamps, (size_t)K * 8 * 8));
    CK(hipMemset(dstamps, 0, (size_t)K * 8 * 8));
#endif
    CK(hipMemcpy(dg, grads.data(), (size_t)K * n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcp, cps.data(), K * sizeof(CutParams), hipMemcpyHostToDevice));
    CK(hipMemset(dpart, 0, (size_t)2 * G * NV * 8));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    std::vector<double> hQ((size_t)n * n);
    auto reset = [&]() {
        std::fill(hQ.begin(), hQ.end(), 0.0);
        for (long long i = 0; i < n; ++i) hQ[(size_t)i * n + i] = 1.0;
        CK(hipMemcpy(dQ, hQ.data(), (size_t)n * n * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dxc, xc0.data(), n * 8, hipMemcpyHostToDevice));
        DevState h{};
        h.kappa = 1.0;
        h.scale = 1.0;
        h.tol = -1.0;
        CK(hipMemcpy(dst, &h, sizeof h, hipMemcpyHostToDevice));
    };
    auto launch = [&](long long first, long long count) {
        ResidentArgs A{};
        A.Q = dQ; A.ld = ld; A.n = n; A.T = T; A.R = R; A.S = S; A.qgrads = dg; A.qparams = dcp; A.qstatus = dstat; A.qtsq = dtsq;
        A.first = first; A.count = count; A.xc = dxc; A.st = dst; A.part = dpart; A.omega_part = dom; A.ctr = dctr;
        A.calc = EllCalcDev::make(n, 1);
        A.stamps = dstamps;
        A.fault_at = -1;   // (the test hook of the failure contract: never)
        CK(hipMemsetAsync(dctr, 0, RS_BAR_WORDS * sizeof(unsigned), st));
        if (R == 1) hipLaunchKernelGGL(k_ell_resident<1>, dim3((unsigned)G), dim3(RS_THREADS), 0, st, A);
        else if (R == 2) hipLaunchKernelGGL(k_ell_resident<2>, dim3((unsigned)G), dim3(RS_THREADS), 0, st, A);
        else hipLaunchKernelGGL(k_ell_resident<3>, dim3((unsigned)G), dim3(RS_THREADS), 0, st, A);
        CK(hipGetLastError());
    };
    // ---- correctness: K0 cuts against the host loop
    reset();
    launch(0, K0);
    CK(hipStreamSynchronize(st));
    std::vector<double> gQ((size_t)n * n), gxc((size_t)n), gtsq((size_t)K0);
    std::vector<int> gstat((size_t)K0);
    DevState hs;
    CK(hipMemcpy(gQ.data(), dQ, (size_t)n * n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gxc.data(), dxc, n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gtsq.data(), dtsq, K0 * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gstat.data(), dstat, K0 * sizeof(int), hipMemcpyDeviceToHost));
    CK(hipMemcpy(&hs, dst, sizeof hs, hipMemcpyDeviceToHost));
    {
        std::vector<double> Q((size_t)n * n, 0.0), xc(xc0), gt((size_t)n);
        for (long long i = 0; i < n; ++i) Q[(size_t)i * n + i] = 1.0;
        double kappa = 1.0, worst_t = 0.0;
        const double nf = (double)n;
        bool ok = true;
        for (int k = 0; k < K0; ++k) {
            const double* g = &grads[(size_t)k * n];
            double omega = 0.0;
            for (long long i = 0; i < n; ++i) {
                double a = 0.0;
                for (long long j = 0; j < n; ++j) a += Q[(size_t)i * n + j] * g[j];
                gt[i] = a;
            }
            for (long long i = 0; i < n; ++i) omega += g[i] * gt[i];
            const double tsq = kappa * omega, beta = cps[k].b0, tau = std::sqrt(tsq);
            // EllCalc::calc_bias_cut (src/ell_calc.rs:870-877, 453-459, 550-553)
            const double eta = tau + nf * beta, rho = eta / (nf + 1.0), sigma = 2.0 * rho / (tau + beta);
            const double delta = (nf * nf / (nf * nf - 1.0)) * (1.0 - (beta / tau) * (beta / tau));
            const double roo = rho / omega, ratio = sigma / omega;
            for (long long i = 0; i < n; ++i) xc[i] -= roo * gt[i];
            for (long long i = 0; i < n; ++i) {
                const double r = ratio * gt[i];
                for (long long j = 0; j <= i; ++j) {
                    Q[(size_t)i * n + j] -= r * gt[j];
                    Q[(size_t)j * n + i] = Q[(size_t)i * n + j];
                }
            }
            kappa *= delta;
            worst_t = std::max(worst_t, std::fabs(gtsq[k] - tsq) / tsq);
            ok = ok && gstat[k] == 0;
        }
        double dq = 0.0, dx = 0.0, sx = 0.0;
        for (long long i = 0; i < n; ++i) {
            for (long long j = 0; j <= i; ++j) dq = std::max(dq, std::fabs(gQ[(size_t)i * n + j] - Q[(size_t)i * n + j]));
            dx = std::max(dx, std::fabs(gxc[i] - xc[i]));
            sx = std::max(sx, std::fabs(xc[i]));
        }
        printf("check after %d cuts: statuses %s, max rel tsq err %.2e, |Q - Q_host| (lower) %.2e, xc rel %.2e, kappa rel %.2e, solve_err %d\n", K0,
               ok ? "ok" : "BAD", worst_t, dq, dx / sx, std::fabs(hs.kappa - kappa) / kappa, hs.solve_err);
    }
    // ---- timing: K cuts per launch (the launch includes parking and writing back the tiles)
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    std::vector<float> ms_k, ms_0;
    for (int r = 0; r < rounds + 1; ++r) {
        reset();
        CK(hipEventRecord(a, st));
        launch(0, K);
        CK(hipEventRecord(b, st));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (r) ms_k.push_back(ms);
        CK(hipEventRecord(a, st));
        launch(0, 0);  // park + write back only
        CK(hipEventRecord(b, st));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
        if (r) ms_0.push_back(ms);
    }
    CK(hipMemcpy(&hs, dst, sizeof hs, hipMemcpyDeviceToHost));
    std::sort(ms_k.begin(), ms_k.end());
    std::sort(ms_0.begin(), ms_0.end());
    const double mk = ms_k[ms_k.size() / 2], m0 = ms_0[ms_0.size() / 2];
    printf("K=%d cuts per launch: %.3f ms (park + write back alone %.3f ms) -> %.2f us per update, %.0f updates/s incl. park/write-back; last status %d solve_err %d\n",
           K, mk, m0, (mk - m0) / K * 1e3, K / (mk * 1e-3), hs.status, hs.solve_err);
#ifdef RS_TIMELINE
    {
        std::vector<unsigned long long> h((size_t)K * 8);
        CK(hipMemcpy(h.data(), dstamps, (size_t)K * 8 * 8, hipMemcpyDeviceToHost));
        const char* names[8] = {"gemv partials + reductions", "sync + partial stores", "grid barrier", "y on own blocks + omega", "EllCalc", "xc + rank-1", "(next cut)", "-"};
        double acc[8] = {0};
        int cnt = 0;
        for (int k = 20; k + 1 < K; ++k) {  // steady state
            for (int p = 0; p < 6; ++p) acc[p] += (double)(h[(size_t)k * 8 + p + 1] - h[(size_t)k * 8 + p]) / 100.0;
            acc[6] += (double)(h[(size_t)(k + 1) * 8] - h[(size_t)k * 8 + 6]) / 100.0;
            ++cnt;
        }
        printf("timeline of workgroup 0 (us per cut, mean over %d cuts):\n", cnt);
        for (int p = 0; p < 7; ++p) printf("  %-26s %7.2f\n", names[p], acc[p] / cnt);
    }
#endif
    return 0;
}
