// lowpass_runner.cpp -- the reference's lowpass tests (src/oracles/lowpass_oracle.rs:169-240,
// tests/stress_tests.rs:7-24) through the C++ host mirror on the MI355X engine, one JSON object per case.
// Three ways of running the same loop must agree: the generic host driver with the device oracle behind the
// OracleOptim interface, the pipelined host driver, and the device-resident loop.
#include <cmath>
#include <cstdio>
#include <string>

#include "../../ellalgo-rs_amd/host/ellhip/lowpass_hip.hpp"

using namespace ellhip;

static void emit(const std::string& name, size_t niter, bool has_x, double gamma, double tsq, const Arr* x) {
    printf("{\"case\": \"%s\", \"niter\": %zu, \"has_x\": %s, \"gamma\": %.17g, \"tsq\": %.17g, \"x\": [", name.c_str(), niter,
           has_x ? "true" : "false", gamma, tsq);
    if (x)
        for (size_t i = 0; i < x->size(); ++i) printf("%s%.17g", i ? ", " : "", (*x)[i]);
    printf("]}\n");
}

static LowpassOracleHip corrected_case(size_t ndim) {
    const double delta1 = 20.0 * std::log10(1.0 + 0.025), delta2 = 20.0 * std::log10(0.125);
    const double lo = std::pow(10.0, -delta1 / 20.0), up = std::pow(10.0, delta1 / 20.0), sp = std::pow(10.0, delta2 / 20.0);
    return LowpassOracleHip(ndim, 0.12, 0.20, lo * lo, up * up, sp * sp);
}

int main() {
    // run_lowpass (:174-185): ndim 32, Ell::new_with_scalar(40, 0), Options{50000, 1e-14}
    {
        auto omega = create_lowpass_case(32);
        EllHip ellip = EllHip::new_with_scalar(40.0, Arr(32, 0.0));
        double sp_sq = omega.sp_sq;
        auto [x, niter] = cutting_plane_optim(omega, ellip, sp_sq, Options(50000, 1e-14));
        emit("run_lowpass_host", niter, x.has_value(), sp_sq, ellip.tsq(), nullptr);
    }
    {
        auto omega = create_lowpass_case(32);
        EllHip ellip = EllHip::new_with_scalar(40.0, Arr(32, 0.0));
        double sp_sq = omega.sp_sq;
        auto [x, niter] = cutting_plane_optim_device(omega, ellip, sp_sq, Options(50000, 1e-14));
        emit("run_lowpass_device", niter, x.has_value(), sp_sq, ellip.tsq(), nullptr);
    }
    // tests/stress_tests.rs:7-24
    {
        auto omega = create_lowpass_case(128);
        EllHip v = EllHip::new_with_scalar(1.0, Arr(128, 0.0));
        double sp_sq = omega.sp_sq;
        auto [x, niter] = cutting_plane_optim_device(omega, v, sp_sq, Options(50000, 1e-14));
        emit("stress_high_dimension_device", niter, x.has_value(), sp_sq, v.tsq(), nullptr);
    }
    {
        auto omega = create_lowpass_case(32);
        EllHip v = EllHip::new_with_scalar(1.0, Arr(32, 0.0));
        double sp_sq = 1e-12;
        auto [x, niter] = cutting_plane_optim_device(omega, v, sp_sq, Options(50000, 1e-14));
        emit("stress_many_iterations_device", niter, x.has_value(), sp_sq, v.tsq(), nullptr);
    }
    // :187-239 (the oracle alone)
    {
        auto oracle = create_lowpass_case(32);
        auto res = oracle.assess_feas(Arr(32, 0.0));
        emit("oracle_zero", 0, res.has_value(), res ? res->second.beta0 : 0.0, res && res->second.beta1 ? *res->second.beta1 : 0.0,
             res ? &res->first : nullptr);
        auto omega = create_lowpass_case(32);
        Arr h(32, 0.0);
        h[0] = 1.0;
        double sp_sq = omega.sp_sq;
        auto cut = omega.assess_optim(h, sp_sq);
        emit("oracle_direct", 0, !cut.first.first.empty(), cut.first.second.beta0, 0.0, nullptr);
    }
    // corrected constants, 200 iterations, three drivers
    for (int mode = 0; mode < 3; ++mode) {
        auto omega = corrected_case(32);
        EllHip ellip = EllHip::new_with_scalar(40.0, Arr(32, 0.0));
        double gamma = omega.sp_sq;
        Options opt(200, 1e-14);
        std::pair<std::optional<Arr>, std::size_t> r;
        if (mode == 0) r = cutting_plane_optim(omega, ellip, gamma, opt);
        if (mode == 1) r = cutting_plane_optim_pipelined(omega, ellip, gamma, opt);
        if (mode == 2) r = cutting_plane_optim_device(omega, ellip, gamma, opt);
        const char* names[3] = {"corrected_host", "corrected_pipelined", "corrected_device"};
        emit(names[mode], r.second, r.first.has_value(), gamma, ellip.tsq(), r.first ? &*r.first : nullptr);
    }
    // feasibility loop
    for (int mode = 0; mode < 2; ++mode) {
        LowpassOracleHip omega(32, 0.12, 0.20, 0.5, 1.5, 0.3);
        EllHip ellip = EllHip::new_with_scalar(40.0, Arr(32, 0.0));
        Options opt(2000, 1e-14);
        auto r = mode == 0 ? cutting_plane_feas(omega, ellip, opt) : cutting_plane_feas_device(omega, ellip, opt);
        emit(mode == 0 ? "feas_host" : "feas_device", r.second, r.first.has_value(), 0.0, ellip.tsq(), r.first ? &*r.first : nullptr);
    }
    return 0;
}
