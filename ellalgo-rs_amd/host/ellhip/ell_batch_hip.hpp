// ell_batch_hip.hpp -- B independent `Ell` search spaces of one dimension n <= 128 behind one handle
// (include/ellhip_batch.h).  The C++ counterpart of a `Vec<Ell>` whose elements are updated together:
// `update_bias_cut(cuts)` is `for b in 0..B { space[b].update_bias_cut(&cuts[b]) }` in one launch, bit-identical to
// the CPU arithmetic.  `from_space` makes B clones of one EllHip (BSearchAdaptor's clone per probe,
// src/cutting_plane.rs:410).
#pragma once

#include <cstdint>
#include <vector>

#include "../../../include/ellhip_batch.h"
#include "ell_hip.hpp"

namespace ellhip {

class EllBatchHip {
  public:
    // Ell::new_with_scalar(val[b], xc[b]) for every b (src/ell.rs:71-73)
    static EllBatchHip new_with_scalar(const Arr& val, const std::vector<Arr>& xc, int device = -1) {
        return EllBatchHip(&val, nullptr, nullptr, xc, device);
    }
    // Ell::new(diag[b], xc[b]) (:55-57)
    static EllBatchHip make(const std::vector<Arr>& diag, const std::vector<Arr>& xc, int device = -1) {
        return EllBatchHip(nullptr, nullptr, &diag, xc, device);
    }
    // Ell::new_with_matrix(kappa[b], mq[b], xc[b]) (:31-41); mq[b] is n*n row-major
    static EllBatchHip new_with_matrix(const Arr& kappa, const std::vector<Arr>& mq, const std::vector<Arr>& xc,
                                       int device = -1) {
        return EllBatchHip(&kappa, &mq, nullptr, xc, device);
    }
    template <int VARIANT>
    static EllBatchHip from_space(SpaceHip<VARIANT>& space, std::size_t B) {
        EllBatchHip r;
        check(ellhip_batch_from_space(&r.h_, space.handle(), (int64_t)B), "ellhip_batch_from_space");
        r.B_ = B;
        r.n_ = space.ndim();
        return r;
    }
    EllBatchHip(const EllBatchHip&) = delete;
    EllBatchHip& operator=(const EllBatchHip&) = delete;
    EllBatchHip(EllBatchHip&& o) noexcept : h_(o.h_), B_(o.B_), n_(o.n_) { o.h_ = nullptr; }
    ~EllBatchHip() { ellhip_batch_destroy(h_); }

    std::size_t size() const { return B_; }
    std::size_t ndim() const { return n_; }

    // one cut per ellipsoid; returns the CutStatus of each
    template <class Cut>
    std::vector<CutStatus> update_bias_cut(const std::vector<std::pair<Arr, Cut>>& cuts) {
        return update(ELLHIP_CUT_BIAS, cuts);
    }
    template <class Cut>
    std::vector<CutStatus> update_central_cut(const std::vector<std::pair<Arr, Cut>>& cuts) {
        return update(ELLHIP_CUT_CENTRAL, cuts);
    }
    template <class Cut>
    std::vector<CutStatus> update_q(const std::vector<std::pair<Arr, Cut>>& cuts) {
        return update(ELLHIP_CUT_Q, cuts);
    }

    std::vector<Arr> xc() const {
        Arr flat(B_ * n_);
        check(ellhip_batch_get_xc(h_, flat.data()), "ellhip_batch_get_xc");
        return split(flat, n_);
    }
    std::vector<Arr> mq() const {
        Arr flat(B_ * n_ * n_);
        check(ellhip_batch_get_mq(h_, flat.data()), "ellhip_batch_get_mq");
        return split(flat, n_ * n_);
    }
    Arr kappa() const {
        Arr k(B_);
        check(ellhip_batch_get_kappa(h_, k.data()), "ellhip_batch_get_kappa");
        return k;
    }
    Arr tsq() const {
        Arr t(B_);
        check(ellhip_batch_get_tsq(h_, t.data()), "ellhip_batch_get_tsq");
        return t;
    }
    void set_no_defer_trick(bool f) { check(ellhip_batch_set_no_defer_trick(h_, f ? 1 : 0), "set_no_defer_trick"); }
    ellhip_batch* handle() { return h_; }

  private:
    EllBatchHip() = default;
    EllBatchHip(const Arr* kappa, const std::vector<Arr>* mq, const std::vector<Arr>* diag, const std::vector<Arr>& xc,
                int device)
        : B_(xc.size()), n_(xc.empty() ? 0 : xc[0].size()) {
        if (kappa && kappa->size() != B_) throw Error(ELLHIP_E_INVALID, "kappa must have B entries");
        const Arr fx = flatten(xc, n_);
        Arr fm, fd;
        if (mq) fm = flatten(*mq, n_ * n_);
        if (diag) fd = flatten(*diag, n_);
        check(ellhip_batch_create(&h_, (int64_t)B_, (int64_t)n_, kappa ? kappa->data() : nullptr,
                                  mq ? fm.data() : nullptr, diag ? fd.data() : nullptr, fx.data(), device),
              "ellhip_batch_create");
    }
    Arr flatten(const std::vector<Arr>& v, std::size_t each) const {
        if (v.size() != B_) throw Error(ELLHIP_E_INVALID, "need one entry per ellipsoid");
        Arr flat;
        flat.reserve(B_ * each);
        for (const Arr& a : v) {
            if (a.size() != each) throw Error(ELLHIP_E_INVALID, "dimension mismatch inside the batch");
            flat.insert(flat.end(), a.begin(), a.end());
        }
        return flat;
    }
    static std::vector<Arr> split(const Arr& flat, std::size_t each) {
        std::vector<Arr> out;
        for (std::size_t o = 0; o < flat.size(); o += each) out.emplace_back(flat.begin() + o, flat.begin() + o + each);
        return out;
    }
    template <class Cut>
    std::vector<CutStatus> update(int kind, const std::vector<std::pair<Arr, Cut>>& cuts) {
        if (cuts.size() != B_) throw Error(ELLHIP_E_INVALID, "need one cut per ellipsoid");
        std::vector<int32_t> kinds(B_, kind), has1(B_), status(B_);
        Arr grads, b0(B_), b1(B_);
        grads.reserve(B_ * n_);
        for (std::size_t b = 0; b < B_; ++b) {
            if (cuts[b].first.size() != n_) throw Error(ELLHIP_E_INVALID, "gradient dimension mismatch");
            grads.insert(grads.end(), cuts[b].first.begin(), cuts[b].first.end());
            const CutScalars c = cut_scalars(cuts[b].second);
            b0[b] = c.beta0;
            has1[b] = c.has_beta1;
            b1[b] = c.beta1;
        }
        check(ellhip_batch_update(h_, 1, kinds.data(), grads.data(), b0.data(), has1.data(), b1.data(), status.data(),
                                  nullptr),
              "ellhip_batch_update");
        std::vector<CutStatus> out(B_);
        for (std::size_t b = 0; b < B_; ++b) out[b] = static_cast<CutStatus>(status[b]);
        return out;
    }

    ellhip_batch* h_ = nullptr;
    std::size_t B_ = 0, n_ = 0;
};

}  // namespace ellhip
