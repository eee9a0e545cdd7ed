// sharded_hip.hpp -- the row-partitioned Ell search space for C++ hosts (include/ellhip_sharded.h).
//
//   ShardedEllHip   one process per GPU; the library issues the one collective of an update through RCCL.  Every rank
//                   constructs it with the same arguments (and the unique id rank 0 made) and makes the same calls.
//   EllShardGroup   ONE process that owns all P row blocks (several GPUs driven from one thread, or several blocks on
//                   one GPU): the exchange is ellhip_shards_exchange (device-to-device copies).
// Both have the SearchSpace surface of src/cutting_plane.rs:154-182 (xc, tsq, update_bias_cut / central_cut / q,
// set_xc), so cutting_plane_feas / optim / optim_q of cutting_plane.hpp run on them unchanged.  Results equal the
// unsharded EllHip bit for bit at depth 1 (same kernels, same summation shapes, tests/cpp/sharded_runner.cpp).
#pragma once

#include <array>
#include <memory>
#include <vector>

#include "../../../include/ellhip_sharded.h"
#include "ell_hip.hpp"

namespace ellhip {

using NcclId = std::array<unsigned char, ELLHIP_NCCL_ID_BYTES>;

inline NcclId make_nccl_id() {  // rank 0; ship the bytes to the other ranks
    NcclId id{};
    check(ellhip_sharded_unique_id(id.data()), "ellhip_sharded_unique_id");
    return id;
}

class ShardedEllHip {
  public:
    // Ell::new_with_scalar (src/ell.rs:71-73) on `nranks` GPUs; `id` from make_nccl_id() (nullptr with one rank)
    static ShardedEllHip new_with_scalar(double val, const Arr& xc, int rank, int nranks, const NcclId* id,
                                         int partition = ELLHIP_SHARD_EQUAL_BLOCKS, int defer_depth = 1, int device = -1) {
        return ShardedEllHip(val, nullptr, nullptr, xc, rank, nranks, id, partition, defer_depth, device);
    }
    // Ell::new (src/ell.rs:55-57)
    static ShardedEllHip make(const Arr& val, const Arr& xc, int rank, int nranks, const NcclId* id,
                              int partition = ELLHIP_SHARD_EQUAL_BLOCKS, int defer_depth = 1, int device = -1) {
        return ShardedEllHip(1.0, nullptr, val.data(), xc, rank, nranks, id, partition, defer_depth, device);
    }
    // Ell::new_with_matrix (src/ell.rs:31-41); mq_rows = THIS rank's rows (see rows())
    static ShardedEllHip new_with_matrix(double kappa, const Arr& mq_rows, const Arr& xc, int rank, int nranks,
                                         const NcclId* id, int partition = ELLHIP_SHARD_EQUAL_BLOCKS,
                                         int defer_depth = 1, int device = -1) {
        return ShardedEllHip(kappa, mq_rows.data(), nullptr, xc, rank, nranks, id, partition, defer_depth, device);
    }
    static std::pair<int64_t, int64_t> rows(int64_t n, int nranks, int rank, int partition = ELLHIP_SHARD_EQUAL_BLOCKS) {
        int64_t r0 = 0, nr = 0;
        check(ellhip_sharded_partition(n, nranks, rank, partition, &r0, &nr), "ellhip_sharded_partition");
        return {r0, nr};
    }
    ShardedEllHip(ShardedEllHip&& o) noexcept : h_(o.h_), n_(o.n_) { o.h_ = nullptr; }
    ShardedEllHip(const ShardedEllHip&) = delete;  // (a collective clone is not part of this boundary)
    ~ShardedEllHip() { ellhip_sharded_destroy(h_); }

    Arr xc() const {
        Arr out(n_);
        check(ellhip_sharded_get_xc(h_, out.data()), "ellhip_sharded_get_xc");
        return out;
    }
    double tsq() const { return ellhip_sharded_tsq(h_); }
    double kappa() const { return ellhip_sharded_kappa(h_); }
    void set_xc(const Arr& x) { check(ellhip_sharded_set_xc(h_, x.data()), "ellhip_sharded_set_xc"); }
    template <class Cut>
    CutStatus update_bias_cut(const std::pair<Arr, Cut>& cut) { return update(ELLHIP_CUT_BIAS, cut); }
    template <class Cut>
    CutStatus update_central_cut(const std::pair<Arr, Cut>& cut) { return update(ELLHIP_CUT_CENTRAL, cut); }
    template <class Cut>
    CutStatus update_q(const std::pair<Arr, Cut>& cut) { return update(ELLHIP_CUT_Q, cut); }
    ellhip_sharded* handle() { return h_; }

  private:
    ShardedEllHip(double kappa, const double* mq_rows, const double* diag, const Arr& xc, int rank, int nranks,
                  const NcclId* id, int partition, int defer_depth, int device)
        : n_(xc.size()) {
        check(ellhip_sharded_create(&h_, (int64_t)n_, kappa, mq_rows, diag, xc.data(), device, rank, nranks,
                                    id ? id->data() : nullptr, nullptr, partition, defer_depth),
              "ellhip_sharded_create");
    }
    template <class Cut>
    CutStatus update(int kind, const std::pair<Arr, Cut>& cut) {
        if (cut.first.size() != n_) throw Error(ELLHIP_E_INVALID, "update: gradient dimension mismatch");
        const CutScalars b = cut_scalars(cut.second);
        return static_cast<CutStatus>(check(
            ellhip_sharded_update(h_, kind, cut.first.data(), b.beta0, b.has_beta1, b.beta1), "ellhip_sharded_update"));
    }
    ellhip_sharded* h_ = nullptr;
    std::size_t n_ = 0;
};

class EllShardGroup {
  public:
    // P equal row blocks of Ell::new_with_scalar(val, xc); devices[r] = the GPU of block r
    static EllShardGroup new_with_scalar(double val, const Arr& xc, const std::vector<int>& devices) {
        return EllShardGroup(val, nullptr, nullptr, xc, devices);
    }
    static EllShardGroup make(const Arr& val, const Arr& xc, const std::vector<int>& devices) {
        return EllShardGroup(1.0, nullptr, val.data(), xc, devices);
    }
    // the whole n x n matrix on the host; every block uploads its own rows
    static EllShardGroup new_with_matrix(double kappa, const Arr& mq, const Arr& xc, const std::vector<int>& devices) {
        if (mq.size() != xc.size() * xc.size()) throw Error(ELLHIP_E_INVALID, "mq must be n*n");
        return EllShardGroup(kappa, mq.data(), nullptr, xc, devices);
    }
    EllShardGroup(EllShardGroup&& o) noexcept : sh_(std::move(o.sh_)), n_(o.n_) { o.sh_.clear(); }
    EllShardGroup(const EllShardGroup&) = delete;
    ~EllShardGroup() {
        for (ellhip_space* s : sh_) ellhip_destroy(s);
    }

    Arr xc() const {  // replicated: any block has it
        Arr out(n_);
        check(ellhip_get_xc(sh_[0], out.data()), "ellhip_get_xc");
        return out;
    }
    double tsq() const { return ellhip_tsq(sh_[0]); }
    double kappa() const { return ellhip_kappa(sh_[0]); }
    void set_xc(const Arr& x) {
        for (ellhip_space* s : sh_) check(ellhip_set_xc(s, x.data()), "ellhip_set_xc");
    }
    void set_defer_depth(int depth) {
        for (ellhip_space* s : sh_) check(ellhip_set_defer_depth(s, depth), "ellhip_set_defer_depth");
    }
    template <class Cut>
    CutStatus update_bias_cut(const std::pair<Arr, Cut>& cut) { return update(ELLHIP_CUT_BIAS, cut); }
    template <class Cut>
    CutStatus update_central_cut(const std::pair<Arr, Cut>& cut) { return update(ELLHIP_CUT_CENTRAL, cut); }
    template <class Cut>
    CutStatus update_q(const std::pair<Arr, Cut>& cut) { return update(ELLHIP_CUT_Q, cut); }
    Arr mq() const {  // all rows, block after block
        Arr out(n_ * n_);
        std::size_t at = 0;
        for (std::size_t r = 0; r < sh_.size(); ++r) {
            check(ellhip_get_mq(sh_[r], out.data() + at), "ellhip_get_mq");
            at += nrows_[r] * n_;
        }
        return out;
    }
    std::size_t nblocks() const { return sh_.size(); }

  private:
    EllShardGroup(double kappa, const double* mq, const double* diag, const Arr& xc, const std::vector<int>& devices)
        : n_(xc.size()) {
        const int P = (int)devices.size();
        if (P < 1) throw Error(ELLHIP_E_INVALID, "EllShardGroup: at least one row block (device) is required");
        try {
            for (int r = 0; r < P; ++r) {
                int64_t r0 = 0, nr = 0;
                check(ellhip_sharded_partition((int64_t)n_, P, r, ELLHIP_SHARD_EQUAL_BLOCKS, &r0, &nr), "ellhip_sharded_partition");
                ellhip_space* s = nullptr;
                check(ellhip_create_shard(&s, (int64_t)n_, r0, nr, kappa, mq ? mq + (std::size_t)r0 * n_ : nullptr, diag,
                                          xc.data(), devices[(std::size_t)r]), "ellhip_create_shard");
                sh_.push_back(s);
                nrows_.push_back((std::size_t)nr);
            }
        } catch (...) {  // (the destructor does not run for a constructor that throws: release the blocks made so far)
            for (ellhip_space* s : sh_) ellhip_destroy(s);
            sh_.clear();
            throw;
        }
    }
    template <class Cut>
    CutStatus update(int kind, const std::pair<Arr, Cut>& cut) {
        if (cut.first.size() != n_) throw Error(ELLHIP_E_INVALID, "update: gradient dimension mismatch");
        const CutScalars b = cut_scalars(cut.second);
        for (ellhip_space* s : sh_)  // phase 1: local passes (asynchronous, one stream per block)
            check(ellhip_update_begin(s, kind, cut.first.data(), b.beta0, b.has_beta1, b.beta1), "ellhip_update_begin");
        check(ellhip_shards_exchange(sh_.data(), (int)sh_.size()), "ellhip_shards_exchange");
        int status = -1;
        for (ellhip_space* s : sh_) {  // phase 2: redundant scalar stage + local shrink; identical status everywhere
            const int st = check(ellhip_update_end(s), "ellhip_update_end");
            if (status >= 0 && st != status) throw Error(ELLHIP_E_STATE, "row blocks disagree on the cut status");
            status = st;
        }
        return static_cast<CutStatus>(status);
    }
    std::vector<ellhip_space*> sh_;
    std::vector<std::size_t> nrows_;
    std::size_t n_ = 0;
};

}  // namespace ellhip
