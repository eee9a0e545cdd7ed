// ell_hip.hpp -- `Ell` / `EllStable` search spaces backed by the MI355X engine (C ABI, include/ellhip.h).
//
// C++ counterpart of the Rust shim in INTEGRATION.md: same constructors and methods as
// src/ell.rs:31-78,140-180 and src/ell_stable.rs:18-35,128-166; the state lives in HBM, every method
// is one C-ABI call.  Copy construction is `Clone` (device-to-device), destruction is `Drop`.
// A library failure (negative return code) throws ellhip::Error; a Rust binding maps it to
// CutStatus::Unknown instead.
#pragma once

#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>

#include "../../../include/ellhip.h"
#include "cutting_plane.hpp"

namespace ellhip {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what + ": " + ellhip_last_error()), code(c) {}
};

inline int check(int rc, const char* what) {
    if (rc < 0) throw Error(rc, what);
    return rc;
}

template <int VARIANT>
class SpaceHip {
  public:
    // new_with_matrix (src/ell.rs:31-41): mq is n*n row-major
    static SpaceHip new_with_matrix(double kappa, const Arr& mq, const Arr& xc, int device = -1) {
        if (mq.size() != xc.size() * xc.size()) throw Error(ELLHIP_E_INVALID, "mq must be n*n");
        return SpaceHip(kappa, mq.data(), nullptr, xc, device);
    }
    // new (src/ell.rs:55-57): diag(val), kappa = 1
    static SpaceHip make(const Arr& val, const Arr& xc, int device = -1) {
        if (val.size() != xc.size()) throw Error(ELLHIP_E_INVALID, "val must have n entries");
        return SpaceHip(1.0, nullptr, val.data(), xc, device);
    }
    // new_with_scalar (src/ell.rs:71-73): identity, kappa = val
    static SpaceHip new_with_scalar(double val, const Arr& xc, int device = -1) {
        return SpaceHip(val, nullptr, nullptr, xc, device);
    }
    // from_covariance (src/ell.rs:76-78)
    static SpaceHip from_covariance(const Arr& cov, const Arr& xc, int device = -1) {
        return new_with_matrix(1.0, cov, xc, device);
    }

    SpaceHip(const SpaceHip& o) : n_(o.n_) { check(ellhip_clone(o.h_, &h_), "ellhip_clone"); }
    SpaceHip(SpaceHip&& o) noexcept : h_(o.h_), n_(o.n_) { o.h_ = nullptr; }
    SpaceHip& operator=(SpaceHip o) noexcept {
        std::swap(h_, o.h_);
        std::swap(n_, o.n_);
        return *this;
    }
    ~SpaceHip() { ellhip_destroy(h_); }

    // ---- SearchSpace
    Arr xc() const {
        Arr out(n_);
        check(ellhip_get_xc(h_, out.data()), "ellhip_get_xc");
        return out;
    }
    double tsq() const { return ellhip_tsq(h_); }
    void set_xc(const Arr& x) {
        if (x.size() != n_) throw Error(ELLHIP_E_INVALID, "set_xc: dimension mismatch");
        check(ellhip_set_xc(h_, x.data()), "ellhip_set_xc");
    }
    template <class Cut>
    CutStatus update_bias_cut(const std::pair<Arr, Cut>& cut) { return update(ELLHIP_CUT_BIAS, cut); }
    template <class Cut>
    CutStatus update_central_cut(const std::pair<Arr, Cut>& cut) { return update(ELLHIP_CUT_CENTRAL, cut); }
    template <class Cut>
    CutStatus update_q(const std::pair<Arr, Cut>& cut) { return update(ELLHIP_CUT_Q, cut); }

    // ---- pipelined form (include/ellhip.h "pipelined update"): one pass over Q per update, same bits
    void prime(const Arr& grad) {
        if (grad.size() != n_) throw Error(ELLHIP_E_INVALID, "prime: gradient dimension mismatch");
        check(ellhip_prime(h_, grad.data()), "ellhip_prime");
    }
    template <class Cut>
    CutStatus cut(int kind, const Cut& c) {
        const CutScalars b = cut_scalars(c);
        return static_cast<CutStatus>(check(ellhip_cut(h_, kind, b.beta0, b.has_beta1, b.beta1), "ellhip_cut"));
    }
    void commit() { check(ellhip_commit(h_, nullptr), "ellhip_commit"); }
    void commit(const Arr& next_grad) {
        if (next_grad.size() != n_) throw Error(ELLHIP_E_INVALID, "commit: gradient dimension mismatch");
        check(ellhip_commit(h_, next_grad.data()), "ellhip_commit");
    }

    // ---- public fields of the reference struct
    double kappa() const { return ellhip_kappa(h_); }
    Arr mq() const {
        Arr out(n_ * n_);
        check(ellhip_get_mq(h_, out.data()), "ellhip_get_mq");
        return out;
    }
    void set_no_defer_trick(bool f) { check(ellhip_set_no_defer_trick(h_, f ? 1 : 0), "ellhip_set_no_defer_trick"); }
    // 1 = rewrite Q at every cut (reference data flow); 8 = record cuts, apply them to Q in batches of 8
    void set_defer_depth(int depth) { check(ellhip_set_defer_depth(h_, depth), "ellhip_set_defer_depth"); }
    int defer_depth() const { return ellhip_defer_depth(h_); }
    void flush() { check(ellhip_flush(h_), "ellhip_flush"); }
    // options (include/ellhip.h, "options": ELLHIP_OPT_LOOKAHEAD, ELLHIP_OPT_RESIDENT, ...)
    void set_option(int key, long long value) { check(ellhip_set_option(h_, key, (int64_t)value), "ellhip_set_option"); }
    long long option(int key) const {
        int64_t v = 0;
        check(ellhip_get_option(h_, key, &v), "ellhip_get_option");
        return (long long)v;
    }
    // a recorded cut sequence replayed from device memory (ellhip_queue_*): `grads` holds k gradients of ndim() each,
    // beta1[i] = NaN for a single cut.  run() takes the pipelined schedule, which looks ahead in the queue where the
    // handle records its updates; results() returns every cut's CutStatus (3 = not run: the
    // queue halted before it, src/cutting_plane.rs:222,308) and tsq.
    void queue_upload(const std::vector<int32_t>& kinds, const Arr& grads, const Arr& beta0, const Arr& beta1) {
        const std::size_t k = kinds.size();
        if (grads.size() != k * n_ || beta0.size() != k || beta1.size() != k)
            throw Error(ELLHIP_E_INVALID, "queue_upload: array sizes do not match");
        std::vector<int32_t> has(k);
        Arr b1(k);
        for (std::size_t i = 0; i < k; ++i) {
            has[i] = beta1[i] == beta1[i] ? 1 : 0;
            b1[i] = has[i] ? beta1[i] : 0.0;
        }
        check(ellhip_queue_upload(h_, (int64_t)k, kinds.data(), grads.data(), beta0.data(), has.data(), b1.data()),
              "ellhip_queue_upload");
        qk_ = k;
    }
    void queue_run(std::size_t first, std::size_t count) {
        check(ellhip_queue_run_fused(h_, (int64_t)first, (int64_t)count), "ellhip_queue_run_fused");
    }
    std::pair<std::vector<int32_t>, Arr> queue_results() {
        std::vector<int32_t> st(qk_);
        Arr ts(qk_);
        check(ellhip_queue_results(h_, st.data(), ts.data()), "ellhip_queue_results");
        return {st, ts};
    }
    long long queue_primed() const { return ellhip_queue_primed(h_); }
    std::size_t ndim() const { return n_; }
    ellhip_space* handle() { return h_; }

  private:
    SpaceHip(double kappa, const double* mq, const double* diag, const Arr& xc, int device) : n_(xc.size()) {
        check(ellhip_create(&h_, VARIANT, (int64_t)n_, kappa, mq, diag, xc.data(), device), "ellhip_create");
    }
    template <class Cut>
    CutStatus update(int kind, const std::pair<Arr, Cut>& cut) {
        if (cut.first.size() != n_) throw Error(ELLHIP_E_INVALID, "update: gradient dimension mismatch");
        const CutScalars b = cut_scalars(cut.second);
        return static_cast<CutStatus>(
            check(ellhip_update(h_, kind, cut.first.data(), b.beta0, b.has_beta1, b.beta1), "ellhip_update"));
    }

    ellhip_space* h_ = nullptr;
    std::size_t n_ = 0;
    std::size_t qk_ = 0;
};

using EllHip = SpaceHip<ELLHIP_SPACE_ELL>;
using EllStableHip = SpaceHip<ELLHIP_SPACE_ELL_STABLE>;

// cutting_plane_optim (src/cutting_plane.rs:286-313) restructured for the pipelined engine: the oracle
// for iteration k+1 is queried between the scalar stage and the shrink of iteration k (the new centre
// is already final there), so the shrink and the next GEMV share one pass over Q.  The sequence of
// oracle calls, cuts, statuses and the returned (x_best, niter) are exactly those of the reference loop.
template <class Oracle, int VARIANT>
std::pair<std::optional<Arr>, std::size_t> cutting_plane_optim_pipelined(Oracle& omega, SpaceHip<VARIANT>& space,
                                                                         double& gamma, const Options& options) {
    std::optional<Arr> x_best;
    if (options.max_iters == 0) return {x_best, 0};
    Arr xc = space.xc();
    auto [cut, shrunk] = omega.assess_optim(xc, gamma);
    space.prime(cut.first);
    for (std::size_t niter = 0; niter < options.max_iters; ++niter) {
        if (shrunk) x_best = xc;  // better gamma obtained at the centre the oracle was queried at
        const CutStatus status = space.cut(shrunk ? ELLHIP_CUT_CENTRAL : ELLHIP_CUT_BIAS, cut.second);
        if (status != CutStatus::Success || space.tsq() < options.tolerance) {
            space.commit();
            return {x_best, niter};
        }
        if (niter + 1 == options.max_iters) break;
        xc = space.xc();                                    // x_{k+1}: final before the shrink
        std::tie(cut, shrunk) = omega.assess_optim(xc, gamma);
        space.commit(cut.first);                            // shrink for cut k + GEMV for cut k+1
    }
    space.commit();
    return {x_best, options.max_iters};
}

// cutting_plane_feas (src/cutting_plane.rs:205-227), pipelined the same way.
template <class Oracle, int VARIANT>
std::pair<std::optional<Arr>, std::size_t> cutting_plane_feas_pipelined(Oracle& omega, SpaceHip<VARIANT>& space,
                                                                        const Options& options) {
    if (options.max_iters == 0) return {std::nullopt, 0};
    Arr xc = space.xc();
    auto cut = omega.assess_feas(xc);
    if (!cut.has_value()) return {xc, 0};
    space.prime(cut->first);
    for (std::size_t niter = 0; niter < options.max_iters; ++niter) {
        const CutStatus status = space.cut(ELLHIP_CUT_BIAS, cut->second);
        if (status != CutStatus::Success || space.tsq() < options.tolerance) {
            space.commit();
            return {std::nullopt, niter};
        }
        if (niter + 1 == options.max_iters) break;
        xc = space.xc();
        cut = omega.assess_feas(xc);
        if (!cut.has_value()) {  // feasible solution obtained (found at the top of iteration niter+1)
            space.commit();
            return {xc, niter + 1};
        }
        space.commit(cut->first);
    }
    space.commit();
    return {std::nullopt, options.max_iters};
}

}  // namespace ellhip
