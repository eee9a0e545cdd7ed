// lowpass_hip.hpp -- `LowpassOracle` (src/oracles/lowpass_oracle.rs:7-167) backed by the device-side
// oracle of include/ellhip_lowpass.h, plus the device-resident forms of the two driver loops it is used
// with.  Same constructor arguments, method names and return shapes as the reference, so it plugs into
// the generic drivers of cutting_plane.hpp (`cutting_plane_optim(omega, space, gamma, options)`) exactly
// like the reference's struct plugs into src/cutting_plane.rs; `cutting_plane_optim_device` /
// `cutting_plane_feas_device` run the same loops without the centre, the gradient or the cut values ever
// leaving HBM.
#pragma once

#include <cmath>
#include <cstdint>
#include <optional>
#include <utility>

#include "../../../include/ellhip_lowpass.h"
#include "ell_hip.hpp"

namespace ellhip {

class LowpassOracleHip {
  public:
    using CutChoice = ParallelCut;                // type CutChoice = ParallelCut (:56,137)
    using Cut = std::pair<Arr, ParallelCut>;      // pub type Cut = (Arr, ParallelCut) (:5)

    // LowpassOracle::new (:23-53)
    LowpassOracleHip(std::size_t ndim, double wpass, double wstop, double lp_sq, double up_sq, double sp_sq,
                     int device = -1)
        : n_(ndim), sp_sq(sp_sq) {
        check(ellhip_lowpass_create(&h_, (int64_t)ndim, wpass, wstop, lp_sq, up_sq, sp_sq, nullptr, device),
              "ellhip_lowpass_create");
    }
    // same, with the caller's own table (row-major 15 ndim x ndim)
    LowpassOracleHip(std::size_t ndim, double wpass, double wstop, double lp_sq, double up_sq, double sp_sq,
                     const Arr& spectrum, int device = -1)
        : n_(ndim), sp_sq(sp_sq) {
        if (spectrum.size() != 15 * ndim * ndim) throw Error(ELLHIP_E_INVALID, "spectrum must be 15n x n");
        check(ellhip_lowpass_create(&h_, (int64_t)ndim, wpass, wstop, lp_sq, up_sq, sp_sq, spectrum.data(), device),
              "ellhip_lowpass_create");
    }
    LowpassOracleHip(const LowpassOracleHip&) = delete;  // the reference's struct is not Clone either
    LowpassOracleHip& operator=(const LowpassOracleHip&) = delete;
    LowpassOracleHip(LowpassOracleHip&& o) noexcept : h_(o.h_), n_(o.n_), sp_sq(o.sp_sq) { o.h_ = nullptr; }
    ~LowpassOracleHip() { ellhip_lowpass_destroy(h_); }

    // impl OracleFeas<Arr> (:55-134)
    std::optional<Cut> assess_feas(const Arr& x) {
        if (x.size() != n_) throw Error(ELLHIP_E_INVALID, "assess_feas: dimension mismatch");
        Arr g(n_);
        double b0 = 0.0, b1 = 0.0;
        int has_b1 = 0;
        const int rc = check(ellhip_lowpass_assess_feas(h_, x.data(), g.data(), &b0, &has_b1, &b1),
                             "ellhip_lowpass_assess_feas");
        if (rc == 0) return std::nullopt;
        return Cut{std::move(g), ParallelCut{b0, has_b1 ? std::optional<double>(b1) : std::nullopt}};
    }
    // impl OracleOptim<Arr> (:136-151)
    std::pair<Cut, bool> assess_optim(const Arr& x, double& gamma) {
        if (x.size() != n_) throw Error(ELLHIP_E_INVALID, "assess_optim: dimension mismatch");
        Arr g(n_);
        double b0 = 0.0, b1 = 0.0;
        int has_b1 = 0, shrunk = 0;
        check(ellhip_lowpass_assess_optim(h_, x.data(), &gamma, g.data(), &b0, &has_b1, &b1, &shrunk),
              "ellhip_lowpass_assess_optim");
        sp_sq = gamma;
        return {Cut{std::move(g), ParallelCut{b0, has_b1 ? std::optional<double>(b1) : std::nullopt}}, shrunk != 0};
    }

    // the struct's public fields (:8-20), read back from the device
    struct Fields {
        bool more_alt;
        int idx1, idx2, idx3, kmax, nwpass, nwstop;
        double fmax, sp_sq;
    };
    Fields fields() const {
        int32_t i[7];
        double d[2];
        check(ellhip_lowpass_state(h_, i, d), "ellhip_lowpass_state");
        return Fields{i[0] != 0, i[1], i[2], i[3], i[4], i[5], i[6], d[0], d[1]};
    }
    std::size_t ndim() const { return n_; }
    ellhip_lowpass* handle() { return h_; }

  private:
    ellhip_lowpass* h_ = nullptr;
    std::size_t n_ = 0;

  public:
    double sp_sq;  // pub sp_sq (:15): the value tests read as `omega.sp_sq` before the first call
};

// create_lowpass_case (:153-167), constants exactly as written there (they give lp_sq > up_sq)
inline LowpassOracleHip create_lowpass_case(std::size_t ndim, int device = -1) {
    const double PI = 3.14159265358979323846264338327950288;
    const double delta0_wpass = 0.025;
    const double delta0_wstop = 0.125;
    const double delta1 = 20.0 * std::log10(delta0_wpass * PI);
    const double delta2 = 20.0 * std::log10(delta0_wstop * PI);
    const double low_pass = std::pow(10.0, -delta1 / 20.0);
    const double up_pass = std::pow(10.0, delta1 / 20.0);
    const double stop_pass = std::pow(10.0, delta2 / 20.0);
    return LowpassOracleHip(ndim, 0.12, 0.20, low_pass * low_pass, up_pass * up_pass, stop_pass * stop_pass, device);
}

// cutting_plane_optim (src/cutting_plane.rs:286-313) with both sides on the device.
template <int VARIANT>
std::pair<std::optional<Arr>, std::size_t> cutting_plane_optim_device(LowpassOracleHip& omega, SpaceHip<VARIANT>& space,
                                                                      double& gamma, const Options& options) {
    Arr x_best(space.ndim());
    int has_best = 0;
    int64_t niter = 0;
    check(ellhip_lowpass_optim(space.handle(), omega.handle(), &gamma, (int64_t)options.max_iters, options.tolerance,
                               x_best.data(), &has_best, &niter),
          "ellhip_lowpass_optim");
    omega.sp_sq = gamma;
    if (!has_best) return {std::nullopt, (std::size_t)niter};
    return {std::move(x_best), (std::size_t)niter};
}

// cutting_plane_feas (src/cutting_plane.rs:205-227) with both sides on the device.
template <int VARIANT>
std::pair<std::optional<Arr>, std::size_t> cutting_plane_feas_device(LowpassOracleHip& omega, SpaceHip<VARIANT>& space,
                                                                     const Options& options) {
    Arr x(space.ndim());
    int feasible = 0;
    int64_t niter = 0;
    check(ellhip_lowpass_feas(space.handle(), omega.handle(), (int64_t)options.max_iters, options.tolerance, x.data(),
                              &feasible, &niter),
          "ellhip_lowpass_feas");
    if (!feasible) return {std::nullopt, (std::size_t)niter};
    return {std::move(x), (std::size_t)niter};
}

}  // namespace ellhip
