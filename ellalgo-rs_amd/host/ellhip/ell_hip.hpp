// ell_hip.hpp -- `Ell` / `EllStable` search spaces backed by the MI355X engine (C ABI, include/ellhip.h).
//
// C++ counterpart of the Rust shim in INTEGRATION.md: same constructors and methods as
// src/ell.rs:31-78,140-180 and src/ell_stable.rs:18-35,128-166; the state lives in HBM, every method
// is one C-ABI call.  Copy construction is `Clone` (device-to-device), destruction is `Drop`.
// A library failure (negative return code) throws ellhip::Error; a Rust binding maps it to
// CutStatus::Unknown instead.
#pragma once

#include <stdexcept>
#include <string>
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

    // ---- public fields of the reference struct
    double kappa() const { return ellhip_kappa(h_); }
    Arr mq() const {
        Arr out(n_ * n_);
        check(ellhip_get_mq(h_, out.data()), "ellhip_get_mq");
        return out;
    }
    void set_no_defer_trick(bool f) { check(ellhip_set_no_defer_trick(h_, f ? 1 : 0), "ellhip_set_no_defer_trick"); }
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
};

using EllHip = SpaceHip<ELLHIP_SPACE_ELL>;
using EllStableHip = SpaceHip<ELLHIP_SPACE_ELL_STABLE>;

}  // namespace ellhip
