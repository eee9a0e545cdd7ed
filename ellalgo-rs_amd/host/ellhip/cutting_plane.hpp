// cutting_plane.hpp -- C++ host mirror of the reference's cut types, plugin interfaces and drivers
// (src/cutting_plane.rs).  The Rust toolchain is not available where this engine is built, so the
// host side above the C ABI is written in C++ with the same names, argument meaning and return
// values; a Rust host keeps its own src/cutting_plane.rs unchanged and only swaps the search space
// (INTEGRATION.md).
//
// Rust generics become templates.  The "traits" are structural:
//
//   SearchSpace  (src/cutting_plane.rs:154-182)
//       Arr       xc() const;
//       double    tsq() const;
//       CutStatus update_bias_cut   (const std::pair<Arr, Cut>&);    Cut = SingleCut | ParallelCut
//       CutStatus update_central_cut(const std::pair<Arr, Cut>&);
//       CutStatus update_q          (const std::pair<Arr, Cut>&);
//       void      set_xc(const Arr&);
//       copy-constructible  (= `Clone`, needed by BSearchAdaptor, :392,410)
//   OracleFeas   (:119-126)   std::optional<std::pair<Arr, Cut>> assess_feas(const Arr& xc);
//                             void update(double gamma);            // optional in Rust, required here
//   OracleOptim  (:129-136)   std::pair<std::pair<Arr, Cut>, bool> assess_optim(const Arr& xc, double& gamma);
//   OracleOptimQ (:139-147)   std::tuple<std::pair<Arr, Cut>, bool, Arr, bool>
//                                 assess_optim_q(const Arr& xc, double& gamma, bool retry);
//   OracleBS     (:150-152)   bool assess_bs(double gamma);
#pragma once

#include <cassert>
#include <cstddef>
#include <optional>
#include <tuple>
#include <utility>
#include <vector>

namespace ellhip {

using Arr = std::vector<double>;  // 1-D Arr (src/arr.rs:12-16 with cols == 0)

// src/cutting_plane.rs:9
struct SingleCut {
    double beta;
};
// src/cutting_plane.rs:18
struct ParallelCut {
    double beta0;
    std::optional<double> beta1;
};

// src/cutting_plane.rs:31-37 (declaration order is the ABI value)
enum class CutStatus : int { Success = 0, NoSoln = 1, NoEffect = 2, Unknown = 3 };

// src/cutting_plane.rs:50-100
struct Options {
    std::size_t max_iters = 2000;
    double tolerance = 1e-20;
    bool verbose = false;  // never read by the reference either
    Options() = default;
    Options(std::size_t mi, double tol) : max_iters(mi), tolerance(tol) {}
};

// Flattening of a cut choice into the C ABI's (beta0, has_beta1, beta1) triple.
struct CutScalars {
    double beta0;
    int has_beta1;
    double beta1;
};
inline CutScalars cut_scalars(const SingleCut& c) { return {c.beta, 0, 0.0}; }
inline CutScalars cut_scalars(const ParallelCut& c) {
    return {c.beta0, c.beta1.has_value() ? 1 : 0, c.beta1.value_or(0.0)};
}

using CInfo = std::pair<bool, std::size_t>;  // src/cutting_plane.rs:102

// src/cutting_plane.rs:205-227
template <class Oracle, class Space>
std::pair<std::optional<Arr>, std::size_t> cutting_plane_feas(Oracle& omega, Space& space, const Options& options) {
    for (std::size_t niter = 0; niter < options.max_iters; ++niter) {
        auto cut = omega.assess_feas(space.xc());
        if (!cut.has_value()) return {space.xc(), niter};  // feasible solution obtained
        const CutStatus status = space.update_bias_cut(*cut);
        if (status != CutStatus::Success || space.tsq() < options.tolerance) return {std::nullopt, niter};
    }
    return {std::nullopt, options.max_iters};
}

// src/cutting_plane.rs:286-313
template <class Oracle, class Space>
std::pair<std::optional<Arr>, std::size_t> cutting_plane_optim(Oracle& omega, Space& space, double& gamma,
                                                               const Options& options) {
    std::optional<Arr> x_best;
    for (std::size_t niter = 0; niter < options.max_iters; ++niter) {
        auto [cut, shrunk] = omega.assess_optim(space.xc(), gamma);
        CutStatus status;
        if (shrunk) {  // better gamma obtained
            x_best = space.xc();
            status = space.update_central_cut(cut);
        } else {
            status = space.update_bias_cut(cut);
        }
        if (status != CutStatus::Success || space.tsq() < options.tolerance) return {x_best, niter};
    }
    return {x_best, options.max_iters};
}

// src/cutting_plane.rs:331-374
template <class Oracle, class Space>
std::pair<std::optional<Arr>, std::size_t> cutting_plane_optim_q(Oracle& omega, Space& space_q, double& gamma,
                                                                 const Options& options) {
    std::optional<Arr> x_best;
    bool retry = false;
    for (std::size_t niter = 0; niter < options.max_iters; ++niter) {
        auto [cut, shrunk, x_q, more_alt] = omega.assess_optim_q(space_q.xc(), gamma, retry);
        if (shrunk) {  // best gamma obtained
            x_best = x_q;
            retry = false;
        }
        const CutStatus status = space_q.update_q(cut);
        switch (status) {
            case CutStatus::Success:
                retry = false;
                break;
            case CutStatus::NoSoln:
                return {x_best, niter};
            case CutStatus::NoEffect:
                if (!more_alt) return {x_best, niter};  // no more alternative cut
                retry = true;
                break;
            default:
                break;
        }
        if (space_q.tsq() < options.tolerance) return {x_best, niter};
    }
    return {x_best, options.max_iters};
}

// src/cutting_plane.rs:376-419
template <class Oracle, class Space>
struct BSearchAdaptor {
    Oracle omega;
    Space space;
    Options options;

    BSearchAdaptor(Oracle o, Space s, Options opt) : omega(std::move(o)), space(std::move(s)), options(opt) {}

    bool assess_bs(double gamma) {
        Space probe(space);  // self.space.clone(): a device-to-device copy for the GPU space
        omega.update(gamma);
        auto [x_feas, niter] = cutting_plane_feas(omega, probe, options);
        (void)niter;
        if (x_feas.has_value()) {
            space.set_xc(*x_feas);
            return true;
        }
        return false;
    }
};

// src/cutting_plane.rs:441-466
template <class Oracle>
CInfo bsearch(Oracle& omega, std::pair<double, double>& intrvl, const Options& options) {
    double lower = intrvl.first, upper = intrvl.second;  // the reference copies out and never writes back
    assert(lower <= upper);
    const double u_orig = upper;
    for (std::size_t niter = 0; niter < options.max_iters; ++niter) {
        const double tau = (upper - lower) / 2.0;
        if (tau < options.tolerance) return {upper != u_orig, niter};
        double gamma = lower;
        gamma += tau;
        if (omega.assess_bs(gamma))
            upper = gamma;  // feasible solution obtained
        else
            lower = gamma;
    }
    return {upper != u_orig, options.max_iters};
}

}  // namespace ellhip
