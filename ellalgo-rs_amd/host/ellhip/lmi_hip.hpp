// lmi_hip.hpp -- `LDLTMgr` (src/oracles/ldlt_mgr.rs), `LMIOracle` (src/oracles/lmi_oracle.rs) and `LMI0Oracle`
// (src/oracles/lmi0_oracle.rs) backed by the device-side implementation of include/ellhip_lmi.h.  Same
// constructor arguments, method names and return shapes as the reference, so they plug into the generic
// drivers of cutting_plane.hpp the way the reference's structs plug into src/cutting_plane.rs.
// A matrix is an `Arr` of m*m doubles, row-major (src/arr.rs:12-16).
#pragma once

#include <cstdint>
#include <optional>
#include <utility>
#include <vector>

#include "../../../include/ellhip_lmi.h"
#include "ell_hip.hpp"

namespace ellhip {

namespace detail {
class LmiHandle {
  public:
    LmiHandle(const std::vector<Arr>* mat_f, const Arr* mat_b, std::size_t m, int device) : m_(m) {
        Arr flat;
        if (mat_f) {
            n_ = mat_f->size();
            flat.reserve(n_ * m * m);
            for (const Arr& f : *mat_f) {
                if (f.size() != m * m) throw Error(ELLHIP_E_INVALID, "every F_k must be m*m");
                flat.insert(flat.end(), f.begin(), f.end());
            }
        }
        if (mat_b && mat_b->size() != m * m) throw Error(ELLHIP_E_INVALID, "B must be m*m");
        check(ellhip_lmi_create(&h_, (int64_t)n_, (int64_t)m, mat_f ? flat.data() : nullptr,
                                mat_b ? mat_b->data() : nullptr, device),
              "ellhip_lmi_create");
    }
    LmiHandle(const LmiHandle&) = delete;
    LmiHandle& operator=(const LmiHandle&) = delete;
    LmiHandle(LmiHandle&& o) noexcept : h_(o.h_), n_(o.n_), m_(o.m_) { o.h_ = nullptr; }
    ~LmiHandle() { ellhip_lmi_destroy(h_); }

    // returns true when a cut was produced
    bool assess(const Arr* x, Arr& g, double& ep) {
        if (n_ > 0 && (!x || x->size() != n_)) throw Error(ELLHIP_E_INVALID, "assess_feas: dimension mismatch");
        g.assign(n_, 0.0);
        return check(ellhip_lmi_assess_feas(h_, x ? x->data() : nullptr, g.data(), &ep), "ellhip_lmi_assess_feas") == 1;
    }
    std::pair<std::size_t, std::size_t> pos() const {
        int64_t p[2];
        check(ellhip_lmi_pos(h_, p), "ellhip_lmi_pos");
        return {(std::size_t)p[0], (std::size_t)p[1]};
    }
    Arr wit() const {
        Arr v(m_);
        check(ellhip_lmi_get_witness(h_, v.data()), "ellhip_lmi_get_witness");
        return v;
    }
    Arr sqrt() const {
        Arr r(m_ * m_);
        check(ellhip_lmi_sqrt(h_, r.data()), "ellhip_lmi_sqrt");
        return r;
    }
    std::size_t n() const { return n_; }
    std::size_t m() const { return m_; }

  private:
    ellhip_lmi* h_ = nullptr;
    std::size_t n_ = 0, m_ = 0;
};
}  // namespace detail

// LDLTMgr: factorize / is_spd / witness / sqrt (ldlt_mgr.rs:22-24, 93-112, 129-140)
class LDLTMgrHip {
  public:
    explicit LDLTMgrHip(std::size_t ndim, int device = -1) : ndim_(ndim), device_(device) {}
    bool factorize(const Arr& mat) {
        h_.emplace(nullptr, &mat, ndim_, device_);
        Arr g;
        return !h_->assess(nullptr, g, ep_);
    }
    bool is_spd() const { return pos().second == 0; }
    std::pair<std::size_t, std::size_t> pos() const { return h_ ? h_->pos() : std::pair<std::size_t, std::size_t>{0, 0}; }
    Arr wit() const { return h_ ? h_->wit() : Arr(ndim_, 0.0); }
    double witness() const {
        if (is_spd()) throw Error(ELLHIP_E_STATE, "witness called on SPD matrix");
        return ep_;
    }
    Arr sqrt() const {
        if (!h_) throw Error(ELLHIP_E_STATE, "sqrt before factorize");
        return h_->sqrt();
    }

  private:
    std::size_t ndim_;
    int device_;
    std::optional<detail::LmiHandle> h_;
    double ep_ = 0.0;
};

// LMIOracle (lmi_oracle.rs:5-45): impl OracleFeas<Arr>, CutChoice = SingleCut
class LMIOracleHip {
  public:
    using CutChoice = SingleCut;
    LMIOracleHip(const std::vector<Arr>& mat_f, const Arr& mat_b, std::size_t m, int device = -1)
        : h_(&mat_f, &mat_b, m, device) {}
    std::optional<std::pair<Arr, SingleCut>> assess_feas(const Arr& xc) {
        Arr g;
        double ep = 0.0;
        if (!h_.assess(&xc, g, ep)) return std::nullopt;
        return std::make_pair(std::move(g), SingleCut{ep});
    }
    void update(double) {}  // OracleFeas::update default (src/cutting_plane.rs:125)
    std::pair<std::size_t, std::size_t> pos() const { return h_.pos(); }

  private:
    detail::LmiHandle h_;
};

// LMI0Oracle (lmi0_oracle.rs:4-35): returns (g, ep) with a bare f64, like the reference
class LMI0OracleHip {
  public:
    LMI0OracleHip(const std::vector<Arr>& mat_f, std::size_t m, int device = -1) : h_(&mat_f, nullptr, m, device) {}
    std::optional<std::pair<Arr, double>> assess_feas(const Arr& x) {
        Arr g;
        double ep = 0.0;
        if (!h_.assess(&x, g, ep)) return std::nullopt;
        return std::make_pair(std::move(g), ep);
    }

  private:
    detail::LmiHandle h_;
};

}  // namespace ellhip
