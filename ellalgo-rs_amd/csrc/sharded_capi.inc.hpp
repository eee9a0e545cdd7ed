// sharded_capi.inc.hpp -- implementation of include/ellhip_sharded.h (included at the end of ellhip_capi.hip).
//
// One process per GPU; this rank's row block is an ordinary shard handle (ellhip_create_shard) and the ONE collective
// of an update is issued by the library on the handle's stream through RCCL, which is opened at run time (dlopen), so
// libellhip.so has no link-time dependency on it.  The orchestration is the one of ellalgo-rs_amd/sharded.py (which
// drives the same shard handle through torch.distributed); see DESIGN.md section 7 for the schedule.
#include "../../include/ellhip_sharded.h"

#include <dlfcn.h>

#include <cmath>
#include <mutex>

// Where the real header is installed, the restated ABI below is checked against it at compile time (nothing of it is
// used otherwise: the library is opened at run time).
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#define ELLHIP_HAVE_RCCL_H 1
#endif

namespace {

// The few RCCL entry points used, with the types of <rccl/rccl.h> restated (ncclComm_t is an opaque pointer,
// ncclUniqueId 128 bytes, ncclDouble = 8, ncclSum = 0, ncclSuccess = 0): nothing of RCCL is needed to build.
struct IdBytes {
    char internal[ELLHIP_NCCL_ID_BYTES];
};
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, /* ncclUniqueId by value */ IdBytes, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
};
constexpr int RCCL_DOUBLE = 8, RCCL_SUM = 0;
#ifdef ELLHIP_HAVE_RCCL_H
static_assert(sizeof(ncclUniqueId) == ELLHIP_NCCL_ID_BYTES && sizeof(IdBytes) == sizeof(ncclUniqueId), "ncclUniqueId is 128 bytes");
static_assert((int)ncclDouble == RCCL_DOUBLE && (int)ncclSum == RCCL_SUM && (int)ncclSuccess == 0, "restated RCCL enums");
static_assert(sizeof(ncclComm_t) == sizeof(void*) && sizeof(ncclResult_t) == sizeof(int) && sizeof(ncclDataType_t) == sizeof(int) &&
                  sizeof(ncclRedOp_t) == sizeof(int), "restated RCCL handle / enum sizes");
// the restated signatures convert to the real ones with the same argument list (enums <-> int, ncclComm_t <-> void*,
// ncclUniqueId <-> IdBytes are layout-compatible by the asserts above)
static_assert(std::is_same<decltype(&ncclAllGather), ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t)>::value, "ncclAllGather");
static_assert(std::is_same<decltype(&ncclAllReduce), ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t)>::value, "ncclAllReduce");
static_assert(std::is_same<decltype(&ncclCommInitRank), ncclResult_t (*)(ncclComm_t*, int, ncclUniqueId, int)>::value, "ncclCommInitRank");
static_assert(std::is_same<decltype(&ncclGetUniqueId), ncclResult_t (*)(ncclUniqueId*)>::value, "ncclGetUniqueId");
static_assert(std::is_same<decltype(&ncclCommDestroy), ncclResult_t (*)(ncclComm_t)>::value, "ncclCommDestroy");
#endif

void rccl_open(Rccl& r) {
    // ELLHIP_RCCL_PATH names the library to use (a non-default install, or a test double); otherwise a copy the process
    // has already mapped (PyTorch-ROCm's) is reused, then the usual names are tried.
    const char* env = getenv("ELLHIP_RCCL_PATH");
    std::string err;
    auto note = [&]() {
        const char* e = dlerror();  // (returns the message ONCE and clears it)
        if (e) err = e;
    };
    if (env && *env) {
        r.lib = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
        if (!r.lib) note();
    } else {
        for (const char* nm : {"librccl.so.1", "librccl.so"}) {
            r.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
            if (r.lib) break;
        }
        for (const char* nm : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (r.lib) break;
            r.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (!r.lib) note();
        }
    }
    if (!r.lib) {
        r.why = std::string("librccl could not be opened: ") + (err.empty() ? "?" : err);
        return;
    }
    auto sym = [&](const char* nm) -> void* {
        void* p = dlsym(r.lib, nm);
        if (!p && r.why.empty()) r.why = std::string("librccl lacks ") + nm;
        return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
}

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { rccl_open(r); });
    return &r;
}

int rccl_fail(const char* what, int code) {
    Rccl* r = rccl();
    char buf[384];
    snprintf(buf, sizeof buf, "%s: %s (%d)", what, (r->GetErrorString ? r->GetErrorString(code) : "?"), code);
    return fail(ELLHIP_E_NORCCL, buf);
}

}  // namespace

struct ellhip_sharded {
    ellhip_space* sh = nullptr;
    int rank = 0, nranks = 1, partition = ELLHIP_SHARD_EQUAL_BLOCKS;
    long long n = 0, row0 = 0, nrows = 0;
    void* comm = nullptr;
    bool own_comm = false;
    // host-supplied collective (ellhip_sharded_create_custom / ellhip_sharded_set_collective): used instead of RCCL
    ellhip_allgather_fn user_allgather = nullptr;
    ellhip_allreduce_fn user_allreduce = nullptr;
    void* user_ctx = nullptr;
    long long qk = 0;
    // group runs of symmetric shards: 0 = not decided yet, 1 = EVERY rank holds the group buffers, -1 = some rank could not
    // allocate them: every rank takes the cut-by-cut schedule (decided once, collectively: shard_group_agree)
    int group_state = 0;
    double* d_agree = nullptr;
};

namespace {

// the collective of one update: in place on the buffer that holds Q*g of the gradient just primed
int shard_exchange(ellhip_sharded* s) {
    double* gt = ellhip_gt_dev(s->sh);
    if (s->user_allgather || s->user_allreduce) {
        DeviceGuard guard(s->sh->device);
        const int urc = (s->partition == ELLHIP_SHARD_SYMMETRIC)
                            ? s->user_allreduce(s->user_ctx, gt, s->n, s->sh->stream)
                            : s->user_allgather(s->user_ctx, gt, s->row0, s->nrows, s->sh->stream);
        if (urc != 0) return fail(ELLHIP_E_NORCCL, "the host-supplied collective reported a failure");
        return 0;
    }
    if (!s->comm) return 0;  // one rank, no communicator: the local pass already produced the whole vector
    Rccl* r = rccl();
    DeviceGuard guard(s->sh->device);
    int rc;
    if (s->partition == ELLHIP_SHARD_SYMMETRIC)
        rc = r->AllReduce(gt, gt, (size_t)s->n, RCCL_DOUBLE, RCCL_SUM, s->comm, s->sh->stream);
    else
        rc = r->AllGather(gt + s->row0, gt, (size_t)s->nrows, RCCL_DOUBLE, s->comm, s->sh->stream);
    if (rc != 0) return rccl_fail(s->partition == ELLHIP_SHARD_SYMMETRIC ? "ncclAllReduce" : "ncclAllGather", rc);
    return 0;
}

// Invariant of this file: whenever the shard handle has a primed gradient (ellhip_queue_primed >= 0, or a direct
// gradient between update_begin and update_end), its vector HAS been exchanged -- every call that runs a GEMV is
// followed by shard_exchange before anything else happens.  So "is cut i primed?" is asked of the shard handle itself
// (a shard whose recorded updates were applied by an observer -- flush, get_mq_rows, a depth change -- has dropped its
// prime, a direct update or a halted queue leaves none): nothing is cached here that could go stale, and a vector that
// is already complete is never exchanged twice (an all-reduce is not idempotent).
bool shard_primed(const ellhip_sharded* s, long long index) { return ellhip_queue_primed(s->sh) == index; }

}  // namespace

extern "C" {

int ellhip_sharded_partition(int64_t n, int nranks, int rank, int partition, int64_t* row0_out, int64_t* nrows_out) {
    if (!row0_out || !nrows_out || n < 1 || nranks < 1 || rank < 0 || rank >= nranks)
        return fail(ELLHIP_E_INVALID, "bad partition arguments");
    if (partition == ELLHIP_SHARD_EQUAL_BLOCKS) {
        if (n % nranks) return fail(ELLHIP_E_INVALID, "equal row blocks: n must be divisible by the number of ranks");
        *nrows_out = n / nranks;
        *row0_out = rank * (n / nranks);
        return 0;
    }
    if (partition != ELLHIP_SHARD_SYMMETRIC) return fail(ELLHIP_E_INVALID, "unknown partition");
    const int64_t align = SYMV_H;
    if (n % align || n / align < nranks)
        return fail(ELLHIP_E_INVALID, "symmetric row shards: n must be a multiple of 64 with at least one strip per rank");
    // boundaries at n sqrt(r / P), rounded (half to even, as ellalgo-rs_amd/sharded.py's round()) to whole strips,
    // strictly increasing, room left for the ranks behind
    const int64_t nstrips = n / align;
    int64_t lo = 0, hi = 0;
    int64_t prev = 0;
    for (int r = 1; r <= nranks; ++r) {
        int64_t b;
        if (r == nranks) {
            b = nstrips;
        } else {
            b = (int64_t)std::nearbyint((double)nstrips * std::sqrt((double)r / (double)nranks));
            b = std::min<int64_t>(std::max<int64_t>(b, prev + 1), nstrips - (nranks - r));
        }
        if (r - 1 == rank) {
            lo = prev;
            hi = b;
        }
        prev = b;
    }
    *row0_out = lo * align;
    *nrows_out = (hi - lo) * align;
    return 0;
}

int ellhip_sharded_unique_id(void* id_out) {
    if (!id_out) return fail(ELLHIP_E_INVALID, "id_out is NULL");
    Rccl* r = rccl();
    if (!r->GetUniqueId) return fail(ELLHIP_E_NORCCL, r->why.c_str());
    const int rc = r->GetUniqueId(id_out);
    if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
    return 0;
}

}  // extern "C"

namespace {
int sharded_create_impl(ellhip_sharded** out, int64_t n, double kappa, const double* mq_rows, const double* diag,
                        const double* xc, int device, int rank, int nranks, const void* nccl_id, void* nccl_comm,
                        int partition, int defer_depth, ellhip_allgather_fn ag, ellhip_allreduce_fn ar, void* ctx) {
    if (!out) return fail(ELLHIP_E_INVALID, "out is NULL");
    *out = nullptr;
    const bool custom = ag || ar;
    if (custom && (!ag || !ar)) return fail(ELLHIP_E_INVALID, "a host-supplied collective needs both callbacks");
    if (nccl_id && nccl_comm) return fail(ELLHIP_E_INVALID, "give either the unique id or a communicator, not both");
    if (nranks > 1 && !nccl_id && !nccl_comm && !custom) return fail(ELLHIP_E_INVALID, "more than one rank needs a communicator");
    if (defer_depth != 1 && defer_depth != 8 && defer_depth != 16 && defer_depth != 24)
        return fail(ELLHIP_E_INVALID, "defer depth must be 1, 8, 16 or 24");
    if (partition == ELLHIP_SHARD_SYMMETRIC && defer_depth == 1)
        return fail(ELLHIP_E_INVALID, "symmetric row shards run the recorded schedule only (depth 8 or 16)");
    if (partition == ELLHIP_SHARD_SYMMETRIC && n < 512)
        return fail(ELLHIP_E_INVALID, "symmetric row shards: the lower-triangle schedule needs n >= 512");
    int64_t row0 = 0, nrows = 0;
    int rc = ellhip_sharded_partition(n, nranks, rank, partition, &row0, &nrows);
    if (rc) return rc;
    ellhip_sharded* s = new (std::nothrow) ellhip_sharded();
    if (!s) return fail(ELLHIP_E_NOMEM, "host allocation failed");
    s->rank = rank;
    s->nranks = nranks;
    s->partition = partition;
    s->n = n;
    s->row0 = row0;
    s->nrows = nrows;
    s->user_allgather = ag;
    s->user_allreduce = ar;
    s->user_ctx = ctx;
    rc = ellhip_create_shard(&s->sh, n, row0, nrows, kappa, mq_rows, diag, xc, device);
    if (!rc && partition == ELLHIP_SHARD_SYMMETRIC) rc = ellhip_set_shard_symmetric(s->sh, 1);
    if (!rc && defer_depth != 1) rc = ellhip_set_defer_depth(s->sh, defer_depth);
    if (!rc && (nccl_id || nccl_comm)) {
        Rccl* r = rccl();
        if (!r->CommInitRank || !r->AllGather || !r->AllReduce) {
            rc = fail(ELLHIP_E_NORCCL, r->why.empty() ? "librccl is incomplete" : r->why.c_str());
        } else if (nccl_comm) {
            s->comm = nccl_comm;
        } else {
            DeviceGuard guard(s->sh->device);
            IdBytes id;
            memcpy(&id, nccl_id, sizeof id);
            const int nrc = r->CommInitRank(&s->comm, nranks, id, rank);
            if (nrc != 0) rc = rccl_fail("ncclCommInitRank", nrc);
            else s->own_comm = true;
        }
    }
    if (rc) {
        ellhip_sharded_destroy(s);
        return rc;
    }
    *out = s;
    return 0;
}
}  // namespace

extern "C" {

int ellhip_sharded_create(ellhip_sharded** out, int64_t n, double kappa, const double* mq_rows, const double* diag,
                          const double* xc, int device, int rank, int nranks, const void* nccl_id, void* nccl_comm,
                          int partition, int defer_depth) {
    return sharded_create_impl(out, n, kappa, mq_rows, diag, xc, device, rank, nranks, nccl_id, nccl_comm, partition,
                               defer_depth, nullptr, nullptr, nullptr);
}

int ellhip_sharded_create_custom(ellhip_sharded** out, int64_t n, double kappa, const double* mq_rows, const double* diag,
                                 const double* xc, int device, int rank, int nranks, int partition, int defer_depth,
                                 ellhip_allgather_fn allgather, ellhip_allreduce_fn allreduce, void* ctx) {
    if (!allgather || !allreduce) return fail(ELLHIP_E_INVALID, "both collective callbacks are required");
    return sharded_create_impl(out, n, kappa, mq_rows, diag, xc, device, rank, nranks, nullptr, nullptr, partition,
                               defer_depth, allgather, allreduce, ctx);
}

int ellhip_sharded_set_collective(ellhip_sharded* s, ellhip_allgather_fn allgather, ellhip_allreduce_fn allreduce, void* ctx) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    if ((allgather == nullptr) != (allreduce == nullptr)) return fail(ELLHIP_E_INVALID, "give both callbacks, or neither");
    if (!allgather && s->nranks > 1 && !s->comm)
        return fail(ELLHIP_E_INVALID, "this handle has no RCCL communicator to fall back to");
    int rc = ellhip_synchronize(s->sh);
    if (rc) return rc;
    s->user_allgather = allgather;
    s->user_allreduce = allreduce;
    s->user_ctx = ctx;
    return 0;
}

void ellhip_sharded_destroy(ellhip_sharded* s) {
    if (!s) return;
    if (s->sh) (void)ellhip_synchronize(s->sh);
    if (s->comm && s->own_comm) {
        Rccl* r = rccl();
        if (r->CommDestroy) (void)r->CommDestroy(s->comm);
    }
    if (s->d_agree) (void)hipFree(s->d_agree);
    if (s->sh) ellhip_destroy(s->sh);
    delete s;
}

int ellhip_sharded_update(ellhip_sharded* s, int kind, const double* grad, double beta0, int has_beta1, double beta1) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    int rc = ellhip_update_begin(s->sh, kind, grad, beta0, has_beta1, beta1);
    if (rc) return rc;
    rc = shard_exchange(s);
    if (rc) return rc;
    return ellhip_update_end(s->sh);
}

double ellhip_sharded_tsq(const ellhip_sharded* s) { return s ? ellhip_tsq(s->sh) : 0.0; }
double ellhip_sharded_kappa(const ellhip_sharded* s) { return s ? ellhip_kappa(s->sh) : 0.0; }
int ellhip_sharded_get_xc(const ellhip_sharded* s, double* xc_out) {
    return s ? ellhip_get_xc(s->sh, xc_out) : fail(ELLHIP_E_INVALID, "NULL handle");
}
int ellhip_sharded_set_xc(ellhip_sharded* s, const double* xc) {
    return s ? ellhip_set_xc(s->sh, xc) : fail(ELLHIP_E_INVALID, "NULL handle");
}
int ellhip_sharded_get_mq_rows(ellhip_sharded* s, double* rows_out) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    return ellhip_get_mq(s->sh, rows_out);
}
int ellhip_sharded_set_defer_depth(ellhip_sharded* s, int depth) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    return ellhip_set_defer_depth(s->sh, depth);
}
int ellhip_sharded_flush(ellhip_sharded* s) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    return ellhip_flush(s->sh);
}

int ellhip_sharded_queue_upload(ellhip_sharded* s, int64_t k, const int32_t* kinds, const double* grads,
                                const double* beta0, const int32_t* has_beta1, const double* beta1) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    const int rc = ellhip_queue_upload(s->sh, k, kinds, grads, beta0, has_beta1, beta1);
    if (rc) return rc;
    s->qk = k;
    return 0;
}

int ellhip_sharded_queue_run(ellhip_sharded* s, int64_t first, int64_t count) {
    if (!s || first < 0 || count < 0 || first + count > s->qk) return fail(ELLHIP_E_INVALID, "queue range");
    for (int64_t i = first; i < first + count; ++i) {
        // cut i may already be primed AND exchanged (a preceding queue_run_fused primes the cut after its last one):
        // begin is then a no-op and the complete vector must not be exchanged again
        const bool ready = shard_primed(s, i);
        int rc = ellhip_queue_begin(s->sh, i);
        if (!rc && !ready) rc = shard_exchange(s);
        if (!rc) rc = ellhip_queue_end(s->sh, i);
        if (rc) return rc;
    }
    return 0;
}

namespace {
// the collective of a GROUP of queued cuts (symmetric shards, ELLHIP_OPT_LOOKAHEAD > 3): the shards' partial products of
// all the group's cuts, `count` = cuts x n doubles, added in place -- one all-reduce where the cut-by-cut schedule has one
// per cut
int shard_group_exchange(void* ctx, double* buf, long long count, hipStream_t stream) {
    ellhip_sharded* s = static_cast<ellhip_sharded*>(ctx);
    if (s->user_allreduce) {
        if (s->user_allreduce(s->user_ctx, buf, count, stream) != 0)
            return fail(ELLHIP_E_NORCCL, "the host-supplied collective reported a failure");
        return 0;
    }
    if (!s->comm) return 0;  // one rank: the local products are the products
    const int rc = rccl()->AllReduce(buf, buf, (size_t)count, RCCL_DOUBLE, RCCL_SUM, s->comm, stream);
    return rc != 0 ? rccl_fail("ncclAllReduce", rc) : 0;
}
}  // namespace

namespace {
// The group schedule issues ONE all-reduce of g n doubles per group, the cut-by-cut schedule one of n doubles per cut: the
// ranks must take the same one or their collectives no longer match.  Whether the group buffers fit is a per-rank fact
// (the partial-sum sets depend on the shard's rows, free memory on the device), so the decision is made collectively, once
// per handle: every rank tries to allocate, the ok flags are summed through the handle's own collective, and unless all
// nranks succeeded everybody frees what it got and runs cut by cut.  Returns 1 (groups), 0 (cut by cut) or an error code.
int shard_group_agree(ellhip_sharded* s) {
    if (s->group_state != 0) return s->group_state > 0 ? 1 : 0;
    const int mrc = multi_setup(s->sh);
    if (mrc < 0) return mrc;
    double ok = (mrc == 0) ? 1.0 : 0.0, sum = ok;
    if (s->nranks > 1) {
        DeviceGuard guard(s->sh->device);
        if (!s->d_agree) HIPCHK(hipMalloc(&s->d_agree, sizeof(double)));
        HIPCHK(hipMemcpyAsync(s->d_agree, &ok, sizeof(double), hipMemcpyHostToDevice, s->sh->stream));
        const int xrc = shard_group_exchange(s, s->d_agree, 1, s->sh->stream);
        if (xrc) return xrc;
        HIPCHK(hipMemcpyAsync(&sum, s->d_agree, sizeof(double), hipMemcpyDeviceToHost, s->sh->stream));
        HIPCHK(hipStreamSynchronize(s->sh->stream));
    }
    const bool all = sum > (double)s->nranks - 0.5;
    if (!all) {
        multi_free(s->sh);
        s->sh->lookahead = 1;
    }
    s->group_state = all ? 1 : -1;
    return all ? 1 : 0;
}
}  // namespace

int ellhip_sharded_queue_run_fused(ellhip_sharded* s, int64_t first, int64_t count) {
    if (!s || first < 0 || count < 0 || first + count > s->qk) return fail(ELLHIP_E_INVALID, "queue range");
    if (count == 0) return 0;
    int rc = 0;
    bool groups = s->partition == ELLHIP_SHARD_SYMMETRIC && multi_shard_ok(s->sh) && (s->user_allreduce || s->comm || s->nranks == 1) &&
                  !(s->user_allgather && !s->user_allreduce);
    if (groups) {
        const int arc = shard_group_agree(s);
        if (arc < 0) return arc;
        groups = arc == 1;
    }
    if (groups) {
        // Symmetric shards look ahead like the unsharded queue run: the products of up to 32 queued
        // cuts in one pass over the local trapezoid, ONE all-reduce of the group's vectors, the group stage on every rank.
        DeviceGuard guard(s->sh->device);
        int64_t i = first;
        if (shard_primed(s, first)) {  // primed and exchanged by an earlier call: that cut by itself
            rc = ellhip_queue_cut(s->sh, first);
            if (!rc) rc = ellhip_queue_commit(s->sh, first, -1);
            if (rc) return rc;
            i += 1;
        }
        if (i < first + count) {
            s->sh->grp_exchange = &shard_group_exchange;
            s->sh->grp_exchange_ctx = s;
            rc = queue_run_multi(s->sh, i, first + count - i);
            s->sh->grp_exchange = nullptr;
            s->sh->grp_exchange_ctx = nullptr;
            if (rc == MULTI_NO_MEMORY)  // (cannot happen: shard_group_agree made sure the buffers exist on every rank)
                return fail(ELLHIP_E_NOMEM, "group run of a row shard: the group buffers are gone");
        }
        return rc;
    }
    // pipelined: one pass over the local rows per cut; the collective follows whichever call ran a GEMV
    if (!shard_primed(s, first)) {
        rc = ellhip_queue_prime(s->sh, first);
        if (!rc) rc = shard_exchange(s);
        if (rc) return rc;
    }
    for (int64_t i = first; i < first + count; ++i) {
        const int64_t nxt = (i + 1 < s->qk) ? i + 1 : -1;
        rc = ellhip_queue_cut(s->sh, i);
        if (!rc) rc = ellhip_queue_commit(s->sh, i, nxt);
        if (!rc && nxt >= 0) rc = shard_exchange(s);
        if (rc) return rc;
    }
    return 0;
}

int ellhip_sharded_queue_results(ellhip_sharded* s, int32_t* status_out, double* tsq_out) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    return ellhip_queue_results(s->sh, status_out, tsq_out);
}

int ellhip_sharded_synchronize(ellhip_sharded* s) {
    return s ? ellhip_synchronize(s->sh) : fail(ELLHIP_E_INVALID, "NULL handle");
}

ellhip_space* ellhip_sharded_local(ellhip_sharded* s) { return s ? s->sh : nullptr; }

int ellhip_shards_exchange(ellhip_space* const* shards, int nshards) {
    if (!shards || nshards < 1) return fail(ELLHIP_E_INVALID, "no shards");
    long long covered = 0;
    for (int a = 0; a < nshards; ++a) {
        const ellhip_space* s = shards[a];
        if (!s || !s->sharded || s->variant != ELLHIP_SPACE_ELL || s->shard_symmetric || s->n != shards[0]->n)
            return fail(ELLHIP_E_INVALID, "ellhip_shards_exchange: equal-block Ell row shards of one matrix expected");
        if (s->row0 != covered) return fail(ELLHIP_E_INVALID, "ellhip_shards_exchange: shards must be given in row order and tile the matrix");
        covered += s->nrows;
    }
    if (covered != shards[0]->n) return fail(ELLHIP_E_INVALID, "ellhip_shards_exchange: the shards do not cover all rows");
    // the local passes must have produced their rows before anybody copies them
    for (int a = 0; a < nshards; ++a) {
        DeviceGuard guard(shards[a]->device);
        HIPCHK(hipStreamSynchronize(shards[a]->stream));
    }
    for (int d = 0; d < nshards; ++d) {
        ellhip_space* dst = shards[d];
        DeviceGuard guard(dst->device);
        double* gt_d = dst->d_gt[dst->cur];
        for (int a = 0; a < nshards; ++a) {
            if (a == d) continue;
            const ellhip_space* src = shards[a];
            const double* gt_s = src->d_gt[src->cur];
            const size_t bytes = (size_t)src->nrows * sizeof(double);
            if (src->device == dst->device)
                HIPCHK(hipMemcpyAsync(gt_d + src->row0, gt_s + src->row0, bytes, hipMemcpyDeviceToDevice, dst->stream));
            else
                HIPCHK(hipMemcpyPeerAsync(gt_d + src->row0, dst->device, gt_s + src->row0, src->device, bytes, dst->stream));
        }
    }
    // a source's slot may be rewritten by its owner's next pass only after every reader has copied it
    for (int d = 0; d < nshards; ++d) {
        DeviceGuard guard(shards[d]->device);
        HIPCHK(hipStreamSynchronize(shards[d]->stream));
    }
    return 0;
}

}  // extern "C"
