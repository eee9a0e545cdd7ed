// sharded_ranks_runner.cpp -- the C-ABI row-partitioned Ell (include/ellhip_sharded.h) with P > 1 RANKS, each rank a
// host thread with its own ellhip_sharded handle on the SAME GPU.  Two transports for the one collective per update:
//   mode "rccl":    the library's own RCCL call path (ncclCommInitRank from a unique id, in-place ncclAllGather /
//                   ncclAllReduce on the handle's stream) against tests/cpp/fake_rccl.cpp, which the library opens
//                   through ELLHIP_RCCL_PATH (set by the caller);
//   mode "custom":  ellhip_sharded_create_custom with host-supplied callbacks (tests/cpp/inproc_collective.hpp).
// Every rank makes the same calls with the same cuts:
//   direct updates [0, 10) -> queue_upload -> queue_run [10, 16) (two passes) -> queue_run_fused [16, 24) -> flush
//   (an observer between two pipelined runs: the prime is dropped and redone) -> queue_run_fused [24, 30) ->
//   queue_run [30, 34) (two-pass right after pipelined: cut 30 is already primed AND exchanged) -> queue_run_fused
//   [34, 40) with cut 36 failing (every rank halts at the same index) -> queue_results.
// Checked per rank against (a) an UNSHARDED handle running the same sequence (equal blocks: with the lower-triangle GEMV
// off, bit for bit incl. this rank's rows of Q; symmetric shards: 1e-12) and (b) the CPU oracle (1e-10).
// Usage: sharded_ranks_runner <rccl|custom> <n> <P> <partition 0|1> <depth> [<rccl|custom> <n> <P> <partition> <depth> ...]
//        -> one JSON line per case, "case": "ranks:<mode>:<n>:<P>:<partition>:<depth>" (all cases of a test session run in
//        ONE process: a process per case spent most of its time starting the HIP runtime)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ellhip_sharded.h"
#include "../../oracle/ell_oracle.h"
#include "inproc_collective.hpp"

namespace {

constexpr int K = 40, KFAIL = 36;

struct Cuts {
    std::vector<int32_t> kinds, has1;
    std::vector<double> grads, b0, b1;
};

Cuts make_cuts(int64_t n) {
    Cuts c;
    c.kinds.resize(K);
    c.has1.resize(K);
    c.b0.resize(K);
    c.b1.resize(K);
    c.grads.resize((size_t)K * n);
    unsigned long long s = 0x9E3779B97F4A7C15ull;
    auto u = [&]() {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        return (double)(s >> 11) / 9007199254740992.0;
    };
    for (int i = 0; i < K; ++i) {
        double nrm = 0.0;
        for (int64_t j = 0; j < n; ++j) {
            const double x = u() - 0.5;
            c.grads[(size_t)i * n + j] = x;
            nrm += x * x;
        }
        nrm = std::sqrt(nrm);
        for (int64_t j = 0; j < n; ++j) c.grads[(size_t)i * n + j] /= nrm;
        // kappa0 = 1, Q0 = I, |g| = 1: tau starts at 1 and shrinks slowly at these sizes
        switch (i % 4) {
            case 0: c.kinds[i] = ELLHIP_CUT_BIAS; c.has1[i] = 0; c.b0[i] = 0.02 * u(); c.b1[i] = 0.0; break;
            case 1: c.kinds[i] = ELLHIP_CUT_CENTRAL; c.has1[i] = 1; c.b0[i] = 0.0; c.b1[i] = 0.05 + 0.1 * u(); break;
            case 2: c.kinds[i] = ELLHIP_CUT_BIAS; c.has1[i] = 1; c.b0[i] = 0.01 * u(); c.b1[i] = c.b0[i] + 0.05 + 0.1 * u(); break;
            default: c.kinds[i] = ELLHIP_CUT_Q; c.has1[i] = 0; c.b0[i] = 0.01 * u(); c.b1[i] = 0.0; break;
        }
    }
    c.kinds[KFAIL] = ELLHIP_CUT_BIAS;
    c.has1[KFAIL] = 0;
    c.b0[KFAIL] = 1e9;  // NoSoln: every driver stops there
    return c;
}

struct Result {
    int rc = 0;
    std::string err;
    std::vector<int32_t> direct_status, qstatus;
    std::vector<double> direct_tsq, qtsq, xc, rows;
    double kappa = 0.0, tsq = 0.0;
    int64_t row0 = 0, nrows = 0;
};

#define RK(expr)                                                                     \
    do {                                                                             \
        const int _rc = (expr);                                                      \
        if (_rc < 0) {                                                               \
            out.rc = _rc;                                                            \
            out.err = std::string(#expr) + ": " + ellhip_last_error();               \
            return;                                                                  \
        }                                                                            \
    } while (0)

struct CustomCtx {
    inproc::Group* g;
    int rank;
};
int cb_allgather(void* ctx, double* vec, int64_t offset, int64_t count, void* stream) {
    CustomCtx* c = static_cast<CustomCtx*>(ctx);
    if (offset != (int64_t)c->rank * count) return 9;
    return inproc::allgather(*c->g, c->rank, vec + offset, vec, (size_t)count, static_cast<hipStream_t>(stream));
}
int cb_allreduce(void* ctx, double* vec, int64_t count, void* stream) {
    CustomCtx* c = static_cast<CustomCtx*>(ctx);
    return inproc::allreduce(*c->g, c->rank, vec, vec, (size_t)count, static_cast<hipStream_t>(stream));
}

// the common call sequence, on any handle type through a small adapter
template <class H>
void sequence(H& h, const Cuts& c, int64_t n, Result& out) {
    out.direct_status.assign(10, -99);
    out.direct_tsq.assign(10, 0.0);
    for (int i = 0; i < 10; ++i) {
        const int st = h.update(c.kinds[i], &c.grads[(size_t)i * n], c.b0[i], c.has1[i], c.b1[i]);
        RK(st);
        out.direct_status[i] = st;
        out.direct_tsq[i] = h.tsq();
    }
    RK(h.queue_upload(K, c.kinds.data(), c.grads.data(), c.b0.data(), c.has1.data(), c.b1.data()));
    RK(h.queue_run(10, 6));
    RK(h.queue_run_fused(16, 8));
    RK(h.flush());
    RK(h.queue_run_fused(24, 6));
    RK(h.queue_run(30, 4));
    RK(h.queue_run_fused(34, 6));
    out.qstatus.assign(K, -99);
    out.qtsq.assign(K, 0.0);
    RK(h.queue_results(out.qstatus.data(), out.qtsq.data()));
    out.kappa = h.kappa();
    out.tsq = h.tsq();
    out.xc.resize((size_t)n);
    RK(h.get_xc(out.xc.data()));
    out.rows.resize((size_t)out.nrows * n);
    RK(h.get_rows(out.rows.data()));
}

struct ShardedH {
    ellhip_sharded* s = nullptr;
    int update(int kind, const double* g, double b0, int h1, double b1) { return ellhip_sharded_update(s, kind, g, b0, h1, b1); }
    double tsq() { return ellhip_sharded_tsq(s); }
    double kappa() { return ellhip_sharded_kappa(s); }
    int queue_upload(int64_t k, const int32_t* kinds, const double* grads, const double* b0, const int32_t* h1, const double* b1) {
        return ellhip_sharded_queue_upload(s, k, kinds, grads, b0, h1, b1);
    }
    int queue_run(int64_t a, int64_t cnt) { return ellhip_sharded_queue_run(s, a, cnt); }
    int queue_run_fused(int64_t a, int64_t cnt) { return ellhip_sharded_queue_run_fused(s, a, cnt); }
    int flush() { return ellhip_sharded_flush(s); }
    int queue_results(int32_t* st, double* ts) { return ellhip_sharded_queue_results(s, st, ts); }
    int get_xc(double* x) { return ellhip_sharded_get_xc(s, x); }
    int get_rows(double* r) { return ellhip_sharded_get_mq_rows(s, r); }
};
struct PlainH {
    ellhip_space* s = nullptr;
    int update(int kind, const double* g, double b0, int h1, double b1) { return ellhip_update(s, kind, g, b0, h1, b1); }
    double tsq() { return ellhip_tsq(s); }
    double kappa() { return ellhip_kappa(s); }
    int queue_upload(int64_t k, const int32_t* kinds, const double* grads, const double* b0, const int32_t* h1, const double* b1) {
        return ellhip_queue_upload(s, k, kinds, grads, b0, h1, b1);
    }
    int queue_run(int64_t a, int64_t cnt) { return ellhip_queue_run(s, a, cnt); }
    int queue_run_fused(int64_t a, int64_t cnt) { return ellhip_queue_run_fused(s, a, cnt); }
    int flush() { return ellhip_flush(s); }
    int queue_results(int32_t* st, double* ts) { return ellhip_queue_results(s, st, ts); }
    int get_xc(double* x) { return ellhip_get_xc(s, x); }
    int get_rows(double* r) { return ellhip_get_mq(s, r); }
};

double rel_inf(const double* a, const double* b, size_t m) {
    double d = 0.0, sc = 0.0;
    for (size_t i = 0; i < m; ++i) {
        const double e = std::fabs(a[i] - b[i]);
        if (e > d || e != e) d = (e != e) ? INFINITY : e;
        if (std::fabs(b[i]) > sc) sc = std::fabs(b[i]);
    }
    return sc > 0.0 ? d / sc : d;
}

}  // namespace

int run_case(const std::string& mode, const int64_t n, const int P, const int partition, const int depth) {
    char tagbuf[96];
    std::snprintf(tagbuf, sizeof tagbuf, "ranks:%s:%lld:%d:%d:%d", mode.c_str(), (long long)n, P, partition, depth);
    const std::string tag = tagbuf;
    const Cuts cuts = make_cuts(n);
    std::vector<double> xc0((size_t)n);
    for (int64_t i = 0; i < n; ++i) xc0[(size_t)i] = 0.001 * (double)(i % 13);

    // ---- the P ranks
    char id[ELLHIP_NCCL_ID_BYTES] = {0};
    if (mode == "rccl" && ellhip_sharded_unique_id(id) != 0) {
        std::printf("{\"case\": \"%s\", \"ok\": false, \"error\": \"unique_id: %s\"}\n", tag.c_str(), ellhip_last_error());
        return 0;
    }
    inproc::Group group(P);
    std::vector<Result> res((size_t)P);
    std::vector<CustomCtx> ctx((size_t)P);
    std::vector<std::thread> th;
    for (int r = 0; r < P; ++r) {
        ctx[(size_t)r] = CustomCtx{&group, r};
        th.emplace_back([&, r]() {
            Result& out = res[(size_t)r];
            RK(ellhip_sharded_partition(n, P, r, partition, &out.row0, &out.nrows));
            ShardedH h;
            if (mode == "rccl")
                RK(ellhip_sharded_create(&h.s, n, 1.0, nullptr, nullptr, xc0.data(), 0, r, P, id, nullptr, partition, depth));
            else
                RK(ellhip_sharded_create_custom(&h.s, n, 1.0, nullptr, nullptr, xc0.data(), 0, r, P, partition, depth, cb_allgather,
                                                cb_allreduce, &ctx[(size_t)r]));
            sequence(h, cuts, n, out);
            ellhip_sharded_destroy(h.s);
        });
    }
    for (auto& t : th) t.join();
    for (int r = 0; r < P; ++r)
        if (res[(size_t)r].rc) {
            std::printf("{\"case\": \"%s\", \"ok\": false, \"rank\": %d, \"error\": \"%s\"}\n", tag.c_str(), r, res[(size_t)r].err.c_str());
            return 0;
        }

    // ---- the unsharded engine on the same sequence
    Result ref;
    {
        Result& out = ref;
        out.nrows = n;
        PlainH h;
        const int rc = ellhip_create(&h.s, ELLHIP_SPACE_ELL, n, 1.0, nullptr, nullptr, xc0.data(), 0);
        if (rc) {
            std::printf("{\"case\": \"%s\", \"ok\": false, \"error\": \"ellhip_create: %s\"}\n", tag.c_str(), ellhip_last_error());
            return 0;
        }
        // an equal-block shard runs full-row GEMVs: so must the reference, for the comparison to be one of bits -- and
        // neither may its queue runs take the resident kernel (its own summation shape)
        int orc = ellhip_set_option(h.s, ELLHIP_OPT_RESIDENT, 0);
        if (!orc) orc = (partition == ELLHIP_SHARD_EQUAL_BLOCKS) ? ellhip_set_option(h.s, ELLHIP_OPT_SYMV, 0)
                                                                : ellhip_set_option(h.s, ELLHIP_OPT_SYMV_MIN_N, 512);
        if (!orc) orc = ellhip_set_defer_depth(h.s, depth);
        int64_t got = -1;
        if (!orc) orc = ellhip_get_option(h.s, partition == ELLHIP_SHARD_EQUAL_BLOCKS ? ELLHIP_OPT_SYMV : ELLHIP_OPT_SYMV_MIN_N, &got);
        if (orc || got != (partition == ELLHIP_SHARD_EQUAL_BLOCKS ? 0 : 512) || ellhip_defer_depth(h.s) != depth) {
            std::printf("{\"case\": \"%s\", \"ok\": false, \"error\": \"reference handle options: %s\"}\n", tag.c_str(), ellhip_last_error());
            return 0;
        }
        [&]() { sequence(h, cuts, n, out); }();
        ellhip_destroy(h.s);
        if (ref.rc) {
            std::printf("{\"case\": \"%s\", \"ok\": false, \"error\": \"reference: %s\"}\n", tag.c_str(), ref.err.c_str());
            return 0;
        }
    }

    // ---- the oracle: every cut up to and including the failing one
    orc_ell* o = orc_ell_new(n, 1.0, nullptr, nullptr, xc0.data());
    std::vector<double> otsq(K, 0.0);
    std::vector<int> ostat(K, -99);
    for (int i = 0; i <= KFAIL; ++i) {
        // (the oracle's row-parallel loop: bit-identical to the reference's loop order -- tests/test_oracle_pins.py ties the
        // three forms together at n = 37, 257, 2048 -- and 20x faster at n = 4096, where the reference order's column-strided
        // mirror stores took 6 s per case)
        ostat[i] = orc_ell_update_rowwise_mt(o, cuts.kinds[i], &cuts.grads[(size_t)i * n], cuts.b0[i], cuts.has1[i], cuts.b1[i]);
        otsq[i] = orc_ell_tsq(o);
    }

    bool ok = true, bits = true;
    double worst_ref = 0.0, worst_orc = 0.0;
    std::string why;
    auto note = [&](bool cond, const char* what, int r) {
        if (!cond && why.empty()) why = std::string(what) + " (rank " + std::to_string(r) + ")";
        ok = ok && cond;
    };
    const bool equal = partition == ELLHIP_SHARD_EQUAL_BLOCKS;
    for (int r = 0; r < P; ++r) {
        const Result& a = res[(size_t)r];
        // statuses: the same everywhere, the failing cut halts every rank at the same index
        note(a.direct_status == ref.direct_status, "direct statuses", r);
        note(a.qstatus == ref.qstatus, "queue statuses", r);
        for (int i = 0; i < 10; ++i) note(a.direct_status[i] == ostat[i], "direct status vs oracle", r);
        for (int i = 10; i <= KFAIL; ++i) note(a.qstatus[i] == ostat[i], "queue status vs oracle", r);
        for (int i = KFAIL + 1; i < K; ++i) note(a.qstatus[i] != ELLHIP_SUCCESS, "cut after the halt ran", r);
        // scalars and the centre
        std::vector<double> ts_a(a.direct_tsq), ts_r(ref.direct_tsq), ts_o(otsq.begin(), otsq.begin() + 10);
        ts_a.insert(ts_a.end(), a.qtsq.begin() + 10, a.qtsq.begin() + KFAIL + 1);
        ts_r.insert(ts_r.end(), ref.qtsq.begin() + 10, ref.qtsq.begin() + KFAIL + 1);
        ts_o.insert(ts_o.end(), otsq.begin() + 10, otsq.begin() + KFAIL + 1);
        for (size_t i = 0; i < ts_a.size(); ++i) {
            worst_ref = std::fmax(worst_ref, std::fabs(ts_a[i] - ts_r[i]) / std::fabs(ts_r[i]));
            worst_orc = std::fmax(worst_orc, std::fabs(ts_a[i] - ts_o[i]) / std::fabs(ts_o[i]));
            bits = bits && ts_a[i] == ts_r[i];
        }
        worst_ref = std::fmax(worst_ref, std::fabs(a.kappa - ref.kappa) / std::fabs(ref.kappa));
        worst_orc = std::fmax(worst_orc, std::fabs(a.kappa - orc_ell_kappa(o)) / std::fabs(orc_ell_kappa(o)));
        bits = bits && a.kappa == ref.kappa && std::memcmp(a.xc.data(), ref.xc.data(), (size_t)n * 8) == 0;
        worst_ref = std::fmax(worst_ref, rel_inf(a.xc.data(), ref.xc.data(), (size_t)n));
        worst_orc = std::fmax(worst_orc, rel_inf(a.xc.data(), orc_ell_xc(o), (size_t)n));
        // this rank's rows of Q (symmetric shards: current up to the diagonal only)
        const double* qr = ref.rows.data() + (size_t)a.row0 * n;
        const double* qo = orc_ell_mq(o) + (size_t)a.row0 * n;
        double dr = 0.0, dor = 0.0, sc = 0.0;
        for (int64_t i = 0; i < a.nrows; ++i) {
            const int64_t cend = equal ? n : a.row0 + i + 1;
            for (int64_t j = 0; j < cend; ++j) {
                const double v = a.rows[(size_t)i * n + j];
                dr = std::fmax(dr, std::fabs(v - qr[(size_t)i * n + j]));
                dor = std::fmax(dor, std::fabs(v - qo[(size_t)i * n + j]));
                sc = std::fmax(sc, std::fabs(qo[(size_t)i * n + j]));
                bits = bits && v == qr[(size_t)i * n + j];
            }
        }
        worst_ref = std::fmax(worst_ref, dr / sc);
        worst_orc = std::fmax(worst_orc, dor / sc);
    }
    note(worst_orc <= 1e-10, "oracle tolerance", -1);
    if (equal) note(bits, "equal blocks are not bit-identical to the unsharded engine", -1);
    else note(worst_ref <= 1e-12, "symmetric shards vs unsharded", -1);
    orc_ell_free(o);
    std::printf("{\"case\": \"%s\", \"ok\": %s, \"mode\": \"%s\", \"n\": %lld, \"P\": %d, \"partition\": %d, \"depth\": %d, \"bit_identical\": %s, "
                "\"vs_unsharded\": %.3e, \"vs_oracle\": %.3e, \"collectives\": %ld, \"why\": \"%s\"}\n",
                tag.c_str(), ok ? "true" : "false", mode.c_str(), (long long)n, P, partition, depth, bits ? "true" : "false", worst_ref, worst_orc,
                mode == "custom" ? group.ncalls : -1L, why.c_str());
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 6 || (argc - 1) % 5 != 0) {
        std::fprintf(stderr, "usage: %s <rccl|custom> <n> <P> <partition> <depth> [... more cases]\n", argv[0]);
        return 2;
    }
    for (int a = 1; a + 4 < argc; a += 5) {
        run_case(argv[a], atoll(argv[a + 1]), atoi(argv[a + 2]), atoi(argv[a + 3]), atoi(argv[a + 4]));
        std::fflush(stdout);
    }
    return 0;
}
