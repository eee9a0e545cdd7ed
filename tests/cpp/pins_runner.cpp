// pins_runner.cpp -- runs the reference's end-to-end known-answer cases through
// the C++ host drivers on one of two backends and prints one JSON object per case:
//   -DBACKEND_ORACLE : search space = CPU oracle  (pins the oracle + the host drivers, no GPU)
//   -DBACKEND_HIP    : search space = MI355X engine through the C ABI
#include <cmath>
#include <cstdio>
#include <limits>
#include <string>
#include <type_traits>

#include "example_oracles.hpp"
#ifdef BACKEND_HIP
#include "../../ellalgo-rs_amd/host/ellhip/ell_hip.hpp"
using EllT = ellhip::EllHip;
using EllStableT = ellhip::EllStableHip;
static const char* BACKEND = "hip";
#else
#include "space_oracle.hpp"
using EllT = testspace::OracleEllSpace;
using EllStableT = testspace::OracleEllStableSpace;
static const char* BACKEND = "oracle";
#endif

using namespace ellhip;
using namespace examples;

static void emit(const std::string& name, size_t niter, bool has_x, const Arr* x = nullptr, double gamma = 0.0,
                 int flag = -1) {
    printf("{\"case\": \"%s\", \"backend\": \"%s\", \"niter\": %zu, \"has_x\": %s, \"flag\": %d, \"gamma\": %.17g, \"x\": [",
           name.c_str(), BACKEND, niter, has_x ? "true" : "false", flag, std::isfinite(gamma) ? gamma : 0.0);
    if (x)
        for (size_t i = 0; i < x->size(); ++i) printf("%s%.17g", i ? ", " : "", (*x)[i]);
    printf("]}\n");
}

static bool g_pipelined = false;  // --pipelined: drive Ell through prime / cut / commit (HIP backend only)
static int g_defer = 1;           // --defer8: deferred shrink (HIP backend only)

template <class Space>
static void apply_defer(Space& space) {
#ifdef BACKEND_HIP
    if constexpr (std::is_same_v<Space, ellhip::EllHip>) {
        if (g_defer != 1) space.set_defer_depth(g_defer);
    }
#else
    (void)space;
#endif
}

template <class Space, class Oracle>
static void run_optim(const std::string& name, Space space, Oracle omega, double gamma, Options opt) {
    apply_defer(space);
#ifdef BACKEND_HIP
    if (g_pipelined) {
        auto [x, niter] = cutting_plane_optim_pipelined(omega, space, gamma, opt);
        emit(name, niter, x.has_value(), x ? &*x : nullptr, gamma);
        return;
    }
#endif
    auto [x, niter] = cutting_plane_optim(omega, space, gamma, opt);
    emit(name, niter, x.has_value(), x ? &*x : nullptr, gamma);
}

template <class Space, class Oracle>
static std::pair<std::optional<Arr>, std::size_t> run_feas(Oracle& omega, Space& space, const Options& opt) {
    apply_defer(space);
#ifdef BACKEND_HIP
    if (g_pipelined) return cutting_plane_feas_pipelined(omega, space, opt);
#endif
    return cutting_plane_feas(omega, space, opt);
}

#ifdef BACKEND_HIP
// --queue-replay: a recorded cut sequence through EllHip::queue_upload / queue_run / queue_results (the pipelined queue
// run with its lookahead) against the same cuts taken one ellhip_update at a time on a second handle
static int queue_replay() {
    const size_t n = 1024, k = 40;
    ellhip_set_default_option(ELLHIP_OPT_SYMV_MIN_N, 512);  // (so that this small handle records its updates and looks ahead)
    ellhip_set_default_option(ELLHIP_OPT_RESIDENT, 0);
    Arr xc0(n, 0.0);
    ellhip::EllHip a = ellhip::EllHip::new_with_scalar(1.0, xc0), b = ellhip::EllHip::new_with_scalar(1.0, xc0);
    std::vector<int32_t> kinds(k, 0);
    Arr grads(k * n), b0(k), b1(k);
    unsigned long long h = 88172645463325252ULL;
    for (size_t i = 0; i < k; ++i) {
        double nrm = 0.0;
        for (size_t j = 0; j < n; ++j) {
            h ^= h << 13, h ^= h >> 7, h ^= h << 17;
            grads[i * n + j] = (double)(h >> 11) / 9007199254740992.0 - 0.5;
            nrm += grads[i * n + j] * grads[i * n + j];
        }
        for (size_t j = 0; j < n; ++j) grads[i * n + j] /= std::sqrt(nrm);
        b0[i] = 0.01;
        b1[i] = (i % 2) ? 0.2 : std::numeric_limits<double>::quiet_NaN();  // parallel cuts alternate with deep cuts
    }
    a.queue_upload(kinds, grads, b0, b1);
    a.queue_run(0, 17);
    a.queue_run(17, k - 17);
    auto [st, ts] = a.queue_results();
    double worst = 0.0;
    bool ok = a.option(ELLHIP_OPT_LOOKAHEAD) == 32 && a.defer_depth() == 24;
    for (size_t i = 0; i < k; ++i) {
        Arr g(grads.begin() + i * n, grads.begin() + (i + 1) * n);
        CutStatus s = (i % 2) ? b.update_bias_cut(std::make_pair(g, ParallelCut{b0[i], std::optional<double>(b1[i])}))
                              : b.update_bias_cut(std::make_pair(g, SingleCut{b0[i]}));
        ok = ok && s == CutStatus::Success && st[i] == 0;
        worst = std::max(worst, std::fabs(ts[i] - b.tsq()) / b.tsq());
    }
    Arr xa = a.xc(), xb = b.xc();
    for (size_t j = 0; j < n; ++j) worst = std::max(worst, std::fabs(xa[j] - xb[j]));
    worst = std::max(worst, std::fabs(a.kappa() - b.kappa()) / b.kappa());
    printf("{\"case\": \"queue_replay\", \"ok\": %s, \"worst\": %.3e}\n", ok ? "true" : "false", worst);
    return 0;
}
#endif

int main(int argc, char** argv) {
    bool with_stable = true;
#ifdef BACKEND_HIP
    for (int i = 1; i < argc; ++i)
        if (std::string(argv[i]) == "--queue-replay") return queue_replay();
#endif
    for (int i = 1; i < argc; ++i) {
        if (std::string(argv[i]) == "--no-stable") with_stable = false;
        if (std::string(argv[i]) == "--pipelined") g_pipelined = true;
        if (std::string(argv[i]) == "--defer8") g_defer = 8;
    }
    const double NEG_INF = -std::numeric_limits<double>::infinity();
    const double INF = std::numeric_limits<double>::infinity();
    Options tol10;  // Options { tolerance: 1e-10, ..Default::default() }
    tol10.tolerance = 1e-10;

    // src/example1.rs:41-75
    run_optim("example1_feasible", EllT::new_with_scalar(10.0, {0.0, 0.0}), Example1{}, NEG_INF, tol10);
    run_optim("example1_infeasible1", EllT::make({10.0, 10.0}, {100.0, 100.0}), Example1{}, NEG_INF, Options{});
    run_optim("example1_infeasible2", EllT::make({10.0, 10.0}, {0.0, 0.0}), Example1{}, 100.0, Options{});
    // src/example1_rr.rs:65-99
    run_optim("example1_rr_feasible", EllT::new_with_scalar(10.0, {0.0, 0.0}), Example1RR{}, NEG_INF, tol10);
    run_optim("example1_rr_infeasible1", EllT::make({10.0, 10.0}, {100.0, 100.0}), Example1RR{}, NEG_INF, Options{});
    run_optim("example1_rr_infeasible2", EllT::make({10.0, 10.0}, {0.0, 0.0}), Example1RR{}, 100.0, Options{});
    // src/example4.rs:68-78
    run_optim("example4_feasible", EllT::new_with_scalar(10.0, {0.0, 0.0}), Example4{}, NEG_INF, tol10);
    // src/quasicvx.rs:61-99
    run_optim("quasicvx_feasible", EllT::make({10.0, 10.0}, {0.0, 0.0}), QuasiCvx{}, 0.0, Options(2000, 1e-8));
    run_optim("quasicvx_infeasible1", EllT::new_with_scalar(10.0, {100.0, 100.0}), QuasiCvx{}, 0.0, Options{});
    run_optim("quasicvx_infeasible2", EllT::make({10.0, 10.0}, {0.0, 0.0}), QuasiCvx{}, 100.0, Options{});
    if (with_stable) {
        // src/quasicvx.rs:101-133
        run_optim("quasicvx_feasible_stable", EllStableT::make({10.0, 10.0}, {0.0, 0.0}), QuasiCvx{}, 0.0,
                  Options(2000, 1e-8));
        run_optim("quasicvx_infeasible1_stable", EllStableT::new_with_scalar(10.0, {100.0, 100.0}), QuasiCvx{}, 0.0,
                  Options{});
        run_optim("quasicvx_infeasible2_stable", EllStableT::make({10.0, 10.0}, {0.0, 0.0}), QuasiCvx{}, 100.0,
                  Options{});
    }
    {  // src/example3.rs:67-85
        Options o8;
        o8.tolerance = 1e-8;
        BSearchAdaptor<Example3, EllT> adaptor(Example3{}, EllT::new_with_scalar(100.0, {0.0, 0.0}), o8);
        std::pair<double, double> intrvl{-100.0, 100.0};
        auto [feasible, niter] = bsearch(adaptor, intrvl, o8);
        Arr x = adaptor.space.xc();
        emit("example3_bsearch", niter, true, &x, 0.0, feasible ? 1 : 0);
    }
    {  // src/oracles/profit_oracle.rs:189-243
        run_optim("profit", EllT::make({100.0, 100.0}, {0.0, 0.0}), Profit(20.0, 40.0, 30.5, {0.1, 0.4}, {10.0, 35.0}),
                  0.0, Options{});
        run_optim("profit_rb", EllT::make({100.0, 100.0}, {0.0, 0.0}),
                  ProfitRb(20.0, 40.0, 30.5, {0.1, 0.4}, {10.0, 35.0}, 0.003, 0.007, 1.0, 1.0, 1.0), 0.0, Options{});
        EllT space = EllT::make({100.0, 100.0}, {0.0, 0.0});
        ProfitQ omega(20.0, 40.0, 30.5, {0.1, 0.4}, {10.0, 35.0});
        double gamma = 0.0;
        auto [x, niter] = cutting_plane_optim_q(omega, space, gamma, Options{});
        emit("profit_q", niter, x.has_value(), x ? &*x : nullptr, gamma);
    }
    {  // tests/cutting_plane_tests.rs:130-188
        {
            EllT s = EllT::new_with_scalar(10.0, {0.0, 0.0});
            FeasXY3 om;
            auto [x, niter] = run_feas(om, s, Options(200, 1e-20));
            emit("cp_feas", niter, x.has_value(), x ? &*x : nullptr);
        }
        {
            EllT s = EllT::new_with_scalar(10.0, {0.0, 0.0});
            AlwaysCutFeas om;
            auto [x, niter] = run_feas(om, s, Options(200, 1e-20));
            emit("cp_feas_no_soln", niter, x.has_value());
        }
        run_optim("cp_optim", EllT::new_with_scalar(10.0, {0.0, 0.0}), OptimBox{}, 0.0, Options(200, 1e-20));
        run_optim("cp_optim_no_soln", EllT::new_with_scalar(10.0, {0.0, 0.0}), OptimBox{}, 100.0, Options(4, 1e-20));
        run_optim("cp_optim_max_iters", EllT::new_with_scalar(10.0, {0.0, 0.0}), AlwaysCutOptim{}, 0.0, Options(5, 1e-20));
        {
            EllT s = EllT::new_with_scalar(10.0, {0.0, 0.0});
            AlwaysCutFeas om;
            auto [x, niter] = run_feas(om, s, Options(5, 1e-20));
            emit("cp_feas_max_iters", niter, x.has_value());
        }
    }
    {  // tests/cutting_plane_tests.rs:272-303
        auto run_q = [&](const char* name, auto omega, double gamma, Options opt) {
            EllT s = EllT::new_with_scalar(10.0, {0.0, 0.0});
            auto [x, niter] = cutting_plane_optim_q(omega, s, gamma, opt);
            emit(name, niter, x.has_value(), x ? &*x : nullptr, gamma);
        };
        run_q("cp_optim_q", OptimBoxQ{}, 0.0, Options(200, 1e-20));
        run_q("cp_optim_q_no_soln", OptimBoxQ{}, 100.0, Options(20, 1e-20));
        run_q("cp_optim_q_no_effect", AlwaysCutOptimQ{}, 0.0, Options(5, 1e-20));
    }
    {  // tests/cutting_plane_tests.rs:309-327
        BSPositive om;
        std::pair<double, double> i1{-100.0, 100.0};
        auto [f1, n1] = bsearch(om, i1, Options(2000, 1e-7));
        emit("bsearch", n1, false, nullptr, 0.0, f1 ? 1 : 0);
        std::pair<double, double> i2{-100.0, -50.0};
        auto [f2, n2] = bsearch(om, i2, Options(20, 1e-20));
        emit("bsearch_no_soln", n2, false, nullptr, 0.0, f2 ? 1 : 0);
    }
    {  // tests/cutting_plane_tests.rs:360-370: BSearchAdaptor over an always-feasible-at-0 oracle.
       // (The reference's MyOracleFeas2 implements OracleFeas; the adaptor supplies assess_bs.)
        BSearchAdaptor<FeasXY3, EllT> adaptor(FeasXY3{}, EllT::new_with_scalar(10.0, {0.0, 0.0}), Options{});
        std::pair<double, double> intrvl{-100.0, 100.0};
        auto [feasible, niter] = bsearch(adaptor, intrvl, Options(2000, 1e-8));
        emit("bsearch_adaptor", niter, false, nullptr, 0.0, feasible ? 1 : 0);
    }
    {  // tests/example2_tests.rs:49-68
        EllT s1 = EllT::new_with_scalar(10.0, {0.0, 0.0});
        Example2 o1;
        auto [x1, n1] = run_feas(o1, s1, Options{});
        emit("example2_feasible", n1, x1.has_value(), x1 ? &*x1 : nullptr);
        EllT s2 = EllT::new_with_scalar(10.0, {100.0, 100.0});
        Example2 o2;
        auto [x2, n2] = run_feas(o2, s2, Options{});
        emit("example2_infeasible", n2, x2.has_value());
    }
    {  // tests/integration_test.rs:85-132 (n = 5) and its n = 16 generalisation (BASELINE config 1)
        for (size_t n : {size_t(5), size_t(16)}) {
            Arr target(n), x0(n, 0.0);
            for (size_t i = 0; i < n; ++i) target[i] = double(i + 1);
            run_optim("quad_n" + std::to_string(n), EllT::new_with_scalar(10.0, x0), QuadTarget{target}, INF,
                      Options(2000, 1e-10));
        }
    }
    {  // benches/ellipsoid.rs:25-46 verbatim: starts at xc = 0 => g = 0 => NaN update, exit at niter 0
        for (size_t n : {size_t(10), size_t(50), size_t(100)}) {
            Arr zero(n, 0.0);
            EllT s = EllT::new_with_scalar(10.0, zero);
            QuadTarget om{zero};
            double gamma = NEG_INF;
#ifdef BACKEND_HIP
            auto [x, niter] = g_pipelined ? cutting_plane_optim_pipelined(om, s, gamma, Options{})
                                          : cutting_plane_optim(om, s, gamma, Options{});
#else
            auto [x, niter] = cutting_plane_optim(om, s, gamma, Options{});
#endif
            Arr k{s.kappa(), s.tsq()};
            emit("bench_degenerate_n" + std::to_string(n), niter, x.has_value(), nullptr, 0.0,
                 std::isnan(s.kappa()) ? 1 : 0);
        }
    }
    return 0;
}
