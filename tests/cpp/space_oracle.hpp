// space_oracle.hpp -- SearchSpace adapters over the CPU oracle (oracle/ell_oracle.h).  TESTS ONLY.
// Lets the host drivers of ellhip/cutting_plane.hpp run against the oracle's Ell / EllStable, which
// pins both the oracle and the drivers with the reference's iteration counts.
#pragma once

#include <utility>

#include "../../ellalgo-rs_amd/host/ellhip/cutting_plane.hpp"
#include "../../oracle/ell_oracle.h"

namespace testspace {

using ellhip::Arr;
using ellhip::CutStatus;

class OracleEllSpace {
  public:
    static OracleEllSpace new_with_scalar(double val, const Arr& xc) {
        return OracleEllSpace(orc_ell_new((int64_t)xc.size(), val, nullptr, nullptr, xc.data()));
    }
    static OracleEllSpace make(const Arr& val, const Arr& xc) {
        return OracleEllSpace(orc_ell_new((int64_t)xc.size(), 1.0, nullptr, val.data(), xc.data()));
    }
    OracleEllSpace(const OracleEllSpace& o) : e_(orc_ell_clone(o.e_)) {}
    OracleEllSpace(OracleEllSpace&& o) noexcept : e_(o.e_) { o.e_ = nullptr; }
    ~OracleEllSpace() { orc_ell_free(e_); }

    Arr xc() const { return Arr(orc_ell_xc(e_), orc_ell_xc(e_) + e_->n); }
    double tsq() const { return orc_ell_tsq(e_); }
    void set_xc(const Arr& x) { for (size_t i = 0; i < x.size(); ++i) orc_ell_xc(e_)[i] = x[i]; }
    template <class Cut> CutStatus update_bias_cut(const std::pair<Arr, Cut>& c) { return upd(ORC_CUT_BIAS, c); }
    template <class Cut> CutStatus update_central_cut(const std::pair<Arr, Cut>& c) { return upd(ORC_CUT_CENTRAL, c); }
    template <class Cut> CutStatus update_q(const std::pair<Arr, Cut>& c) { return upd(ORC_CUT_Q, c); }
    double kappa() const { return orc_ell_kappa(e_); }

  private:
    explicit OracleEllSpace(orc_ell* e) : e_(e) {}
    template <class Cut> CutStatus upd(int kind, const std::pair<Arr, Cut>& c) {
        const ellhip::CutScalars b = ellhip::cut_scalars(c.second);
        return static_cast<CutStatus>(orc_ell_update(e_, kind, c.first.data(), b.beta0, b.has_beta1, b.beta1));
    }
    orc_ell* e_;
};

class OracleEllStableSpace {
  public:
    static OracleEllStableSpace new_with_scalar(double val, const Arr& xc) {
        return OracleEllStableSpace(orc_ellstable_new((int64_t)xc.size(), val, nullptr, nullptr, xc.data()));
    }
    static OracleEllStableSpace make(const Arr& val, const Arr& xc) {
        return OracleEllStableSpace(orc_ellstable_new((int64_t)xc.size(), 1.0, nullptr, val.data(), xc.data()));
    }
    OracleEllStableSpace(const OracleEllStableSpace& o) : e_(orc_ellstable_clone(o.e_)) {}
    OracleEllStableSpace(OracleEllStableSpace&& o) noexcept : e_(o.e_) { o.e_ = nullptr; }
    ~OracleEllStableSpace() { orc_ellstable_free(e_); }

    Arr xc() const { return Arr(orc_ellstable_xc(e_), orc_ellstable_xc(e_) + e_->n); }
    double tsq() const { return orc_ellstable_tsq(e_); }
    void set_xc(const Arr& x) { for (size_t i = 0; i < x.size(); ++i) orc_ellstable_xc(e_)[i] = x[i]; }
    template <class Cut> CutStatus update_bias_cut(const std::pair<Arr, Cut>& c) { return upd(ORC_CUT_BIAS, c); }
    template <class Cut> CutStatus update_central_cut(const std::pair<Arr, Cut>& c) { return upd(ORC_CUT_CENTRAL, c); }
    template <class Cut> CutStatus update_q(const std::pair<Arr, Cut>& c) { return upd(ORC_CUT_Q, c); }
    double kappa() const { return orc_ellstable_kappa(e_); }

  private:
    explicit OracleEllStableSpace(orc_ellstable* e) : e_(e) {}
    template <class Cut> CutStatus upd(int kind, const std::pair<Arr, Cut>& c) {
        const ellhip::CutScalars b = ellhip::cut_scalars(c.second);
        return static_cast<CutStatus>(orc_ellstable_update(e_, kind, c.first.data(), b.beta0, b.has_beta1, b.beta1));
    }
    orc_ellstable* e_;
};

}  // namespace testspace
