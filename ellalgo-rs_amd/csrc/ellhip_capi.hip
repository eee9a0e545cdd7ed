// ellhip_capi.hip -- implementation of the C ABI declared in include/ellhip.h.
//
// Host-side runtime of the engine: owns the device buffers of one search space, stages the caller's
// host vectors through pinned memory, issues the kernels of ell_kernels.hpp / ellstable_kernels.hpp
// on one HIP stream and reads the scalar state back.  No torch types, no CPU compute path: if there
// is no HIP device every entry point that needs one fails (ELLHIP_E_NODEVICE).
//
// Every update is built from three primitives (see include/ellhip.h, "pipelined update"):
//   prime   GEMV pass            gt[slot] = Q * g
//   cut     scalar stage         omega, tsq, EllCalc, xc, kappa   (EllStable: the whole update)
//   commit  rank-1 pass, optionally fused with the GEMV pass of the next gradient
// ellhip_update = prime + cut + commit(NULL); the two-phase multi-GPU form splits it after prime.
#include "../../include/ellhip.h"

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "ell_kernels.hpp"
#include "ellstable_kernels.hpp"
#include "resident_kernels.hpp"
#include "group_kernels.hpp"

using namespace ellhip;

namespace {

thread_local std::string g_last_error = "";

int fail(int code, const char* what, hipError_t e = hipSuccess) {
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof buf, "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
    else
        snprintf(buf, sizeof buf, "%s", what);
    g_last_error = buf;
    return code;
}

#define HIPCHK(expr)                                                \
    do {                                                            \
        hipError_t _e = (expr);                                     \
        if (_e != hipSuccess) return fail(ELLHIP_E_HIP, #expr, _e); \
    } while (0)

// hipMemset on device memory is enqueued on the null stream and may return before the fill has run, and every handle's stream is
// non-blocking: a kernel launched right afterwards on the handle's stream can run BEFORE the fill and be overwritten by it (seen once in
// 3600 state-machine walks: the first queued cut's tsq read back as 0 -- the fill of ellhip_queue_upload had landed behind the cut).
// Fills are therefore waited for: the same null-stream hipMemset as before, then a wait for the null stream.  (Moving the fills to the
// handle's own stream changed what the host threads of the multi-rank tests wait for and two suite runs in a row lost a rank case;
// this form changes nothing but the missing wait.)
hipError_t fill_now(void* p, int value, size_t bytes, hipStream_t /*the handle's stream: documentation only*/) {
    const hipError_t e = hipMemset(p, value, bytes);
    return e == hipSuccess ? hipStreamSynchronize(nullptr) : e;
}

struct ProfEvent {
    hipEvent_t a, b;
    int cls;
};

// What a NEW handle starts with (ellhip_set_default_option; process-wide, not thread-safe by contract).  The values are
// the measured best on MI355X; nothing here is read from the environment.
struct Defaults {
    int auto_defer = 1;        // ELLHIP_OPT_AUTO_DEFER
    int symv = 1;              // ELLHIP_OPT_SYMV
    long long symv_min_n = 5120;  // ELLHIP_OPT_SYMV_MIN_N
    int apply_lower = 1;       // ELLHIP_OPT_APPLY_LOWER
    int apply_kernel = -1;     // ELLHIP_OPT_APPLY_KERNEL (-1: by depth)
    int fuse_dots = 1;         // ELLHIP_OPT_FUSE_DOTS
    int resident = 1;          // ELLHIP_OPT_RESIDENT
    int overlap = 1;           // ELLHIP_OPT_OVERLAP
    int lookahead = 32;        // ELLHIP_OPT_LOOKAHEAD
    int queue_depth = 48;      // ELLHIP_OPT_QUEUE_DEPTH
    int stable_solve = 3;      // ELLHIP_OPT_STABLE_SOLVE
    int stable_factor = 2;     // ELLHIP_OPT_STABLE_FACTOR
    int pad = -1;              // ELLHIP_OPT_PAD: extra doubles per row of Q; -1 = by size (create_impl)
    int lp_grid = 0;           // ELLHIP_OPT_LP_GRID: workgroups per LowpassOracle scan launch; 0 = by size
    int lp_wide = -1;          // ELLHIP_OPT_LP_WIDE: column-split scan kernel; -1 = by size
    int batch_threads = 0;     // ELLHIP_OPT_BATCH_THREADS: threads per workgroup of the batched engine; 0 = by size
    int stage_direct = 1;      // ELLHIP_OPT_STAGE_DIRECT: on large-BAR systems the host writes gradients straight into device memory
};
Defaults g_defaults;

enum : int { CLS_GEMV = 0, CLS_SCALAR = 1, CLS_RANK1 = 2, CLS_ST_FWD = 3, CLS_ST_BWD = 4, CLS_ST_FACTOR = 5, CLS_FUSED = 6, CLS_APPLY = 7, CLS_APPLY_GEMV = 8, CLS_SYMV = 9, CLS_SYMV_REDUCE = 10, CLS_LP_SCAN = 11, CLS_LP_FINAL = 12, CLS_RESIDENT = 13 };

struct Shape {
    int rw = 0, unr = 0, nt = 0;
};

}  // namespace

struct ellhip_space {
    int variant = ELLHIP_SPACE_ELL;
    long long n = 0, ld = 0, row0 = 0, nrows = 0;
    int device = 0;
    bool sharded = false;

    double* d_Q = nullptr;           // nrows * ld
    double* d_xc = nullptr;          // n
    double* d_stage[2] = {nullptr, nullptr};   // gradient of slot 0 / 1 (n doubles)
    bool stage_direct = false;       // d_stage is fine-grained device memory the HOST writes through the PCIe BAR (large-BAR systems)
    double* d_gt_own[2] = {nullptr, nullptr};  // Q*g of slot 0 / 1 (n doubles)
    double* d_gt[2] = {nullptr, nullptr};      // buffers in use (own or caller's)
    double* d_work = nullptr;        // EllStable vectors: w, z, gg, q, beta2
    double* d_hpart = nullptr;       // EllStable forward solve with helper workgroups: the helpers' hand-over buffer (n)
    // EllStable kernel forms (ellhip_set_option: ELLHIP_OPT_STABLE_SOLVE / ELLHIP_OPT_STABLE_FACTOR)
    int stable_solve = 3;            // 0: one launch per block; 1: persistent solves; 2: persistent + helper workgroups; 3: 2 on the mirrored layout
    int stable_factor = 2;           // 0: tile kernel that reads the scratch triangle; 1: row kernel from U alone, beside the
                                     // backward solve; 2: factor tiles pulled inside the helped backward solve's launch
    int* d_fnext = nullptr;          // queue position of the factor tiles (reset by k_st_post before every launch)
    int* d_ftiles16 = nullptr;       // k_st_bwd_factor_helped: 16-row tiles (row group << 4 | segment)
    int nftiles16 = 0;
    double* d_qhpart = nullptr;      // ... and the backward helpers' hand-over buffer (n)
    int persist_cap_h = 0;           // resident-workgroup limit of the helped solves (CU count x occupancy of k_st_fwd_helped)
    // EllStable, ELLHIP_OPT_STABLE_SOLVE = 3: the mirrored layout (ellstable_kernels.hpp, StPend)
    bool st_mirrored = false;        // host view: the handle's solves run on the mirrored layout (the device's own flag, StPend.mirrored,
                                     // says whether the lower triangle really holds the mirrored factor: not after a halted queue)
    StPend* d_stpend = nullptr;
    double* d_rbuf = nullptr;        // [2][n] running row scales of the factor (two buffers: StPend says who reads which)
    double* d_wkeep = nullptr;       // [n] w of the last forward solve that really ran (what the scratch triangle is made of)
    int st_since_enter = 0;          // updates issued since the layout was entered (re-entered every ST_MIRROR_PERIOD)
    bool st_prev_persist = true;     // the previous update ran the persistent / helped solves: its k_st_post re-armed their
    bool st_prev_helped = true;      // hand-over buffers for this one (true at creation: alloc_common arms them all)
    double* d_partial = nullptr;     // per-workgroup partial sums of omega (64)
    double* d_pend = nullptr;        // deferred mode: MAXPEND pending gt vectors (n each)
    double* d_cpend = nullptr;       // deferred mode: their coefficients sigma/omega
    double* d_rowpart = nullptr;     // symmetric GEMV: per-segment row partial sums  [n/SYMV_SEG][n]
    double* d_colpart = nullptr;     // symmetric GEMV: per-strip column partial sums [n/SYMV_H][n]
    // pipelined queue runs on the lower-triangle schedule (ELLHIP_OPT_OVERLAP): the next cut's GEMV is issued on a second
    // stream beside this cut's reduction + scalar stage, into the other of two sets of partial sums
    int overlap = 1;
    double* d_rowpart2 = nullptr;
    double* d_colpart2 = nullptr;
    int part_set = 0;                // the set the primed gradient's GEMV wrote / writes (see rowpart_of)
    hipStream_t symv_stream = nullptr;
    // ... and with the GEMVs of up to `lookahead` consecutive queued cuts computed in ONE pass over Q_base (k_symv_multi,
    // ELLHIP_OPT_LOOKAHEAD): vector l of a group writes partial-sum set 2 + l (slices of one allocation)
    int lookahead = 32;
    int queue_depth = 48;            // recorded updates a queue run on the group stage lets pile up before an apply pass (0: the handle's depth)
    double* d_rowpart_m = nullptr;   // [MULTI_MAX][nsegs][n]
    double* d_colpart_m = nullptr;   // [MULTI_MAX][nstrips][n]
    double* d_gT = nullptr;          // [n][16]: a group's gradients side by side (operand layout of k_symm_mfma)
    SymmTile* d_symm_tiles = nullptr;  // k_symm_mfma_q's tiles, largest first
    unsigned* d_symm_queue = nullptr;  // its two counters (one per half of the partial-sum sets), 128 bytes apart
    int symm_ntiles = 0, symm_wgs = 0;
    // ... and the group's scalar stage (group_kernels.hpp)
    double* d_grpY = nullptr;        // [GRP_MAX][n]: y_l = Q_base g_l of the group's cuts
    double* d_gpart = nullptr;       // [GRP_MAX][ceil(n/128)][MAXPEND + 1]
    double* d_cpart = nullptr;       // [ceil(n/128)][GRP_MAX * GRP_MAX]
    double* d_gsums = nullptr;       // [GRP_MAX][MAXPEND + 1] + [GRP_MAX][GRP_MAX]: their sums over the blocks
    GroupOut* d_gout = nullptr;
    // a symmetric row shard's group runs: the owner's all-reduce of the group's partial products (count doubles at buf, in
    // place, on `stream`); set by sharded_capi.inc.hpp around its call of queue_run_multi
    int (*grp_exchange)(void* ctx, double* buf, long long count, hipStream_t stream) = nullptr;
    void* grp_exchange_ctx = nullptr;
    hipEvent_t ev_symv = nullptr;    // side_mark: what was issued on the second stream has finished
    hipEvent_t ev_side_go = nullptr; // side_fork: everything enqueued on the handle's stream up to the fork
    bool side_busy = false;          // work issued on the second stream has not been joined yet (side_fork .. side_join)
    int symv = 1;                    // allow the lower-triangle GEMV in deferred mode (ELLHIP_SYMV=0 disables)
    int symv_rw = 2;
    long long symv_min_n = 5120;     // below this the full-row pass wins for synchronous updates (tools/midsize_sweep.py)
    bool shard_symmetric = false;    // row shard whose GEMVs are partial symmetric sums (ellhip_set_shard_symmetric)
    int symv_seg = SYMV_SEG;         // segment width of the lower-triangle GEMV's tiles (see symv_alloc)
    int apply_lower = 1;             // with symv: apply passes touch the lower triangle only (ELLHIP_APPLY_LOWER)
    int fuse_dots = 1;               // unsharded lower-triangle schedule: k_symv_reduce also yields the scalar stage's dot products (ELLHIP_FUSE_DOTS)
    // queue runs with the matrix parked on-chip (resident_kernels.hpp, ELLHIP_OPT_RESIDENT)
    int resident = 1;
    int rs_R = 0, rs_S = 0;          // super-tile edge / rows chosen for this n and device (0: does not fit)
    double* d_rs_part = nullptr;     // [2][grid][2 R 64]
    double* d_rs_omega = nullptr;    // [2][grid]
    unsigned* d_rs_bar = nullptr;    // RS_BAR_WORDS
    double* d_rs_xc0 = nullptr;      // xc at the start of the batch in flight (restored when the batch is abandoned)
    DevState* d_rs_st0 = nullptr;    // ... and the scalar state
    long long rs_fault_at = -1;      // ELLHIP_OPT_RESIDENT_FAULT (test hook)
    long long rs_abandoned = 0;      // batches abandoned and rerun on the streamed schedule (ELLHIP_OPT_RESIDENT_ABANDONED)
    int dots_np = 0;                 // > 0: d_partial holds dot products of the primed gradient for this depth: [ceil(n/128)][dots_np + 1]
                                     // from k_symv_reduce, or [scalar_groups(n)][...] WITHOUT the g.y column from k_sweep_gemv_dots
    bool dots_need_gy = false;       // the latter: k_scalar_apply_def forms g.y itself
    int apply_kernel = -1;           // lower-triangle apply pass: 2 = k_apply_mfma, 1 = k_apply_lower, 0 = k_sweep_apply<LOWER> (depth 8); -1 = 2 at depth 24, else 1
    bool upper_stale = false;        // strict upper triangle of Q is out of date (see flush_pending)
    int defer = 1;                   // 1 = shrink Q at every cut; MAXPEND = record and apply in batches
    int npend = 0;                   // updates recorded since the last flush (host view, optimistic in queue mode)
    int* d_flags = nullptr;          // EllStable persistent solves: block-ready flags (forward | backward)
    int epoch = 0;                   // hand-off epoch, bumped per persistent launch
    int persist_cap1 = 0;            // workgroups of the persistent solves the device holds at once (CU count x occupancy)
    DevState* d_st = nullptr;

    double* h_stage[2] = {nullptr, nullptr};  // pinned, n doubles each
    DevState* h_result = nullptr;             // pinned
    // live updates (ellhip_update / ellhip_cut): k_publish writes the scalar state and the centre into these right behind
    // the scalar stage and the host polls the sequence number -- no device-to-host copy, no wait for the whole stream
    LiveMirror* h_live = nullptr;             // pinned, fine-grained
    double* h_xc = nullptr;                   // pinned, fine-grained: the centre as of the last published update
    unsigned long long live_seq = 0;
    unsigned* d_pub_arrived = nullptr;        // k_publish: workgroups that have pushed their slice of the centre
    bool xc_host_valid = false;               // h_xc equals d_xc (nothing has written the centre on the device since)
    bool live_arm = false;                    // the do_cut in progress belongs to a live update: the scalar stage may publish itself
    bool live_done = false;                   // ... and did (live_publish has nothing left to launch)

    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t aux_stream = nullptr;           // EllStable: factor update runs beside the persistent backward solve
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;

    int no_defer_trick = 0;
    int use_parallel_cut = 1;
    bool needs_mirror = false;  // caller-supplied non-symmetric matrix, no successful update yet

    // pipeline state
    int cur = 0;                  // slot of the primed / current gradient
    bool primed = false;          // gt[cur] = Q * g[cur] is valid (or in flight)
    const double* g_cur = nullptr;  // device pointer of the primed gradient (stage slot or queue entry)
    bool shrink_pending = false;  // a cut succeeded (or may have: queue mode) and Q has not been shrunk yet
    bool in_two_phase = false;    // update_begin issued, update_end not yet
    CutParams two_phase_cp{};     // cut scalars kept between begin and end
    int dir = 0;                  // direction of the next pass over Q (serpentine)
    long long primed_qindex = -1; // queue index the primed gradient belongs to (-1: a direct gradient)

    // cached scalars (refreshed by every synchronous read-back; `scalars_stale`: queue cuts have been issued since)
    double kappa = 1.0, tsq = 0.0;
    bool scalars_stale = false;

    // device-resident cut queue
    long long qk = 0;
    CutParams* d_qparams = nullptr;
    double* d_qgrads = nullptr;
    int* d_qstatus = nullptr;
    double* d_qtsq = nullptr;

    // per-kernel event timing
    bool profile = false;
    std::vector<ProfEvent> prof_events;
    size_t prof_used = 0;
    double prof_ms[ELLHIP_NKERNEL_CLASSES] = {0};
    long long prof_cnt[ELLHIP_NKERNEL_CLASSES] = {0};

    // launch shapes per role (tunable through the environment for experiments)
    Shape sh_gemv, sh_rank1, sh_fused, sh_apply;
};

namespace {

struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

// ---- profiling helpers ---------------------------------------------------------------------
int prof_flush(ellhip_space* s) {
    for (size_t i = 0; i < s->prof_used; ++i) {
        ProfEvent& pe = s->prof_events[i];
        HIPCHK(hipEventSynchronize(pe.b));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, pe.a, pe.b));
        s->prof_ms[pe.cls] += ms;
        s->prof_cnt[pe.cls] += 1;
    }
    s->prof_used = 0;
    return 0;
}

struct ProfScope {
    ellhip_space* s;
    ProfEvent* pe = nullptr;
    hipStream_t stream;
    ProfScope(ellhip_space* sp, int cls, hipStream_t on = nullptr) : s(sp), stream(on ? on : sp->stream) {
        if (!s->profile) return;
        if (s->prof_used == s->prof_events.size()) {
            if (s->prof_events.size() >= 8192) {
                if (prof_flush(s) != 0) return;
            } else {
                ProfEvent e;
                if (hipEventCreate(&e.a) != hipSuccess) return;
                if (hipEventCreate(&e.b) != hipSuccess) {
                    (void)hipEventDestroy(e.a);
                    return;
                }
                e.cls = cls;
                s->prof_events.push_back(e);
            }
        }
        pe = &s->prof_events[s->prof_used++];
        pe->cls = cls;
        (void)hipEventRecord(pe->a, stream);
    }
    ~ProfScope() {
        if (pe) (void)hipEventRecord(pe->b, stream);
    }
};

// ---- launch-shape selection ------------------------------------------------------------------
// A workgroup covers RW rows; each lane keeps RW*UNR 16-byte loads in flight.  Defaults come from
// in-process A/B sweeps on MI355X (tools/tune_ell.hip, numbers in DESIGN.md / profiles/):
//   * local Q block fits the 256 MiB Infinity Cache (n = 4096: 128 MiB): keep the default cache
//     policy so the next pass finds Q on-die;
//   * larger: the Q stream is touched once per pass, so use the non-temporal policy (GEMV pass 4.9 ->
//     7.0 TB/s at n = 16384).
void pick_shape(ellhip_space* s) {
    const double q_bytes = (double)s->nrows * (double)s->ld * 8.0;
    const double MiB = 1024.0 * 1024.0;
    if (q_bytes <= 200.0 * MiB) {          // lives in the Infinity Cache between passes
        s->sh_gemv = {4, 2, 0};
        s->sh_rank1 = {4, 4, 0};
        s->sh_fused = {4, 4, 0};
    } else if (q_bytes <= 1024.0 * MiB) {  // a few x the cache: read-only pass streams, RMW passes keep what fits
        s->sh_gemv = {4, 4, 1};
        s->sh_rank1 = {4, 4, 0};
        s->sh_fused = {4, 4, 0};
    } else {                               // pure streaming
        s->sh_gemv = {4, 4, 1};
        s->sh_rank1 = {2, 8, 1};
        s->sh_fused = {2, 8, 1};
    }
    s->sh_apply = {4, 1, s->sh_fused.nt};  // 8 pending vectors per column step: more rows per workgroup amortise them
    s->symv = g_defaults.symv;
    s->symv_min_n = g_defaults.symv_min_n;
    s->apply_lower = g_defaults.apply_lower;
    s->apply_kernel = g_defaults.apply_kernel;
    s->fuse_dots = g_defaults.fuse_dots;
    s->overlap = g_defaults.overlap;
    s->lookahead = g_defaults.lookahead;
    s->queue_depth = g_defaults.queue_depth;
    s->resident = g_defaults.resident;
    s->stable_solve = g_defaults.stable_solve;
    s->stable_factor = g_defaults.stable_factor;
}

// One pass over Q in the given roles.  gt_r1: vector of the rank-1 role; gvec / gv_out: operand and
// result of the GEMV role (gv_out is the full-length buffer; the shard's rows start at row0).
template <int VEC, bool NT, bool R1, bool GV, bool SCALE>
int launch_sweep_t(ellhip_space* s, const Shape& sh, const double* gt_r1, const double* gvec, double* gv_out) {
    const long long nr = s->nrows;
    const unsigned grid = (unsigned)((nr + sh.rw - 1) / sh.rw);
    double* out = gv_out ? gv_out + s->row0 : nullptr;
#define SWEEP_CASE(RW, UNR)                                                                                  \
    if (sh.rw == RW && sh.unr == UNR) {                                                                      \
        hipLaunchKernelGGL((k_sweep<RW, UNR, VEC, NT, R1, GV, SCALE>), dim3(grid), dim3(256), 0, s->stream,  \
                           (const double*)s->d_Q, s->d_Q, s->ld, s->n, nr, s->row0, gt_r1, gvec, out, s->d_st, \
                           s->dir);                                                                          \
        return 0;                                                                                            \
    }
    SWEEP_CASE(1, 4) SWEEP_CASE(1, 8) SWEEP_CASE(2, 4) SWEEP_CASE(2, 8) SWEEP_CASE(4, 2) SWEEP_CASE(4, 4)
    SWEEP_CASE(8, 1) SWEEP_CASE(8, 2)
#undef SWEEP_CASE
    return fail(ELLHIP_E_INVALID, "unsupported RW/UNR launch shape (supported: 1x4 1x8 2x4 2x8 4x2 4x4 8x1 8x2)");
}

template <bool R1, bool GV>
int launch_sweep(ellhip_space* s, const Shape& sh, const double* gt_r1, const double* gvec, double* gv_out) {
    const bool even = (s->n % 2) == 0;  // odd n: rows are not 16-byte aligned, use 8-byte accesses
    const bool nt = even && sh.nt;
    const bool scale = R1 && s->no_defer_trick;
    int rc;
    if (!even)
        rc = scale ? launch_sweep_t<1, false, R1, GV, R1>(s, sh, gt_r1, gvec, gv_out)
                   : launch_sweep_t<1, false, R1, GV, false>(s, sh, gt_r1, gvec, gv_out);
    else if (nt)
        rc = scale ? launch_sweep_t<2, true, R1, GV, R1>(s, sh, gt_r1, gvec, gv_out)
                   : launch_sweep_t<2, true, R1, GV, false>(s, sh, gt_r1, gvec, gv_out);
    else
        rc = scale ? launch_sweep_t<2, false, R1, GV, R1>(s, sh, gt_r1, gvec, gv_out)
                   : launch_sweep_t<2, false, R1, GV, false>(s, sh, gt_r1, gvec, gv_out);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    s->dir ^= 1;  // the next pass over Q runs the other way
    return 0;
}

// Full-row GEMV of the deferred schedule with the v_j . g dot products computed beside it (k_sweep_gemv_dots).
template <int VEC, bool NT>
int launch_gemv_dots_t(ellhip_space* s, const Shape& sh, const double* gvec, double* gv_out) {
    const long long nr = s->nrows;
    const unsigned ntiles = (unsigned)((nr + sh.rw - 1) / sh.rw);
    const unsigned grid = ntiles + (unsigned)scalar_groups(s->n);
#define DOTS_CASE(RW, UNR)                                                                                          \
    if (sh.rw == RW && sh.unr == UNR) {                                                                             \
        hipLaunchKernelGGL((k_sweep_gemv_dots<RW, UNR, VEC, NT, 8>), dim3(grid), dim3(256), 0, s->stream,           \
                           (const double*)s->d_Q, s->ld, s->n, nr, s->row0, gvec, gv_out + s->row0, s->d_st, s->dir, \
                           ntiles, (const double*)s->d_pend, s->d_partial);                                         \
        return 0;                                                                                                   \
    }
    DOTS_CASE(1, 4) DOTS_CASE(1, 8) DOTS_CASE(2, 4) DOTS_CASE(2, 8) DOTS_CASE(4, 2) DOTS_CASE(4, 4)
    DOTS_CASE(8, 1) DOTS_CASE(8, 2)
#undef DOTS_CASE
    return fail(ELLHIP_E_INVALID, "unsupported RW/UNR launch shape (supported: 1x4 1x8 2x4 2x8 4x2 4x4 8x1 8x2)");
}

int launch_gemv_dots(ellhip_space* s, const double* gvec, double* gv_out) {
    const bool even = (s->n % 2) == 0;
    const bool nt = even && s->sh_gemv.nt;
    int rc = !even ? launch_gemv_dots_t<1, false>(s, s->sh_gemv, gvec, gv_out)
                   : (nt ? launch_gemv_dots_t<2, true>(s, s->sh_gemv, gvec, gv_out)
                         : launch_gemv_dots_t<2, false>(s, s->sh_gemv, gvec, gv_out));
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    s->dir ^= 1;
    s->dots_np = 8;
    s->dots_need_gy = true;
    return 0;
}

// Deferred mode is in force for Ell with depth > 1, except while the reference semantics need Q itself
// to be rewritten at once (no_defer_trick scaling; the one-off mirror of a non-symmetric input).
bool deferring(const ellhip_space* s) {
    return s->variant == ELLHIP_SPACE_ELL && s->defer > 1 && !s->no_defer_trick && !s->needs_mirror;
}

template <int VEC, bool NT, bool GV, bool LOWER = false>
int launch_apply_t(ellhip_space* s, const double* gvec, double* gv_out) {
    const Shape& sh = s->sh_apply;
    const long long nr = s->nrows;
    const unsigned grid = (unsigned)((nr + sh.rw - 1) / sh.rw);
    double* out = gv_out ? gv_out + s->row0 : nullptr;
    const int dir = LOWER ? 1 : s->dir;  // lower-only: last (longest) rows first, so the short ones fill the tail
#define APPLY_CASE(RW, UNR)                                                                                   \
    if (sh.rw == RW && sh.unr == UNR) {                                                                       \
        hipLaunchKernelGGL((k_sweep_apply<RW, UNR, VEC, NT, GV, LOWER>), dim3(grid), dim3(256), 0, s->stream, \
                           (const double*)s->d_Q, s->d_Q, s->ld, s->n, nr, s->row0, (const double*)s->d_pend, \
                           (const double*)s->d_cpend, gvec, out, s->d_st, dir);                               \
        return 0;                                                                                             \
    }
    APPLY_CASE(1, 2) APPLY_CASE(1, 4) APPLY_CASE(2, 1) APPLY_CASE(2, 2) APPLY_CASE(2, 4) APPLY_CASE(4, 1) APPLY_CASE(4, 2)
#undef APPLY_CASE
    return fail(ELLHIP_E_INVALID, "unsupported ELLHIP_APPLY_RW/UNR shape (supported: 1x2 1x4 2x1 2x2 2x4 4x1 4x2)");
}

bool symv_ok(const ellhip_space* s);

// Apply every pending update to Q (one pass), optionally fused with the GEMV of `gvec`; then clear the slots.
// While the GEMVs of this handle read the lower triangle only (symv_ok), so does the apply pass: half the
// traffic; the strict upper triangle goes stale until make_q_current() mirrors it back.
int side_join(ellhip_space* s);
int flush_pending(ellhip_space* s, const double* gvec, double* gv_out) {
    {
        int jrc = side_join(s);  // a writer of Q: nothing issued ahead on the second stream may still be reading it
        if (jrc) return jrc;
        ProfScope ps(s, gvec ? CLS_APPLY_GEMV : CLS_APPLY);
        const bool even = (s->n % 2) == 0;
        const bool nt = even && s->sh_apply.nt;
        int rc;
        if (!gvec && s->apply_lower && symv_ok(s)) {
            const unsigned grid = (unsigned)((s->nrows + APL_TR - 1) / APL_TR);
#define APL_GO(NPV, NTV)                                                                                        \
    hipLaunchKernelGGL((k_apply_lower<NPV, NTV>), dim3(grid), dim3(256), 0, s->stream, s->d_Q, s->ld, s->n, s->nrows, \
                       s->row0, (const double*)s->d_pend, (const double*)s->d_cpend, (const DevState*)s->d_st)
            // measured at n = 16384, ms per pass: k_apply_mfma 0.43-0.44 at any depth; k_apply_lower 0.40 at depth 8 / 16,
            // 0.62 at depth 24 (its registers); k_sweep_apply<LOWER> 0.44 (depth 8)
            // (more recorded than the handle's depth: a queue run on the group stage let them pile up to 48, queue_run_multi)
            const bool deep = s->npend > s->defer;
            const int apply_kernel = deep ? 2 : (s->apply_kernel < 0 ? (s->defer == 24 ? 2 : 1) : s->apply_kernel);
            if (apply_kernel == 2) {  // the rank-NP update on the FP64 matrix cores (k_apply_mfma)
                const dim3 g2((unsigned)((s->nrows + APM_ROWS - 1) / APM_ROWS), (unsigned)((s->n + APM_COLS - 1) / APM_COLS));
#define APM_GO(NPV, NTV)                                                                                              \
    hipLaunchKernelGGL((k_apply_mfma<NPV, NTV>), g2, dim3(256), 0, s->stream, s->d_Q, s->ld, s->n, s->nrows, s->row0, \
                       (const double*)s->d_pend, (const double*)s->d_cpend, (const DevState*)s->d_st)
                if (deep) { if (nt) APM_GO(48, true); else APM_GO(48, false); }
                else if (s->defer == 24) { if (nt) APM_GO(24, true); else APM_GO(24, false); }
                else if (s->defer == 16) { if (nt) APM_GO(16, true); else APM_GO(16, false); }
                else { if (nt) APM_GO(8, true); else APM_GO(8, false); }
#undef APM_GO
            } else if (s->defer == 24) {  // 24 / 16 pending updates: the 16-row-tile kernel (coefficients in LDS)
                if (nt) APL_GO(24, true); else APL_GO(24, false);
            } else if (s->defer == 16) {
                if (nt) APL_GO(16, true); else APL_GO(16, false);
            } else if (apply_kernel == 1) {  // depth 8, same kernel (0.40 ms; ELLHIP_APPLY_KERNEL=0: k_sweep_apply, 0.44)
                if (nt) APL_GO(8, true); else APL_GO(8, false);
            } else {
                rc = nt ? launch_apply_t<2, true, false, true>(s, nullptr, nullptr)
                        : launch_apply_t<2, false, false, true>(s, nullptr, nullptr);
                if (rc) return rc;
            }
#undef APL_GO
            rc = 0;
            s->upper_stale = true;
        } else if (s->defer != 8) {
            return fail(ELLHIP_E_STATE, "defer depth 16 / 24 needs the lower-triangle schedule");
        } else if (gvec)
            rc = !even ? launch_apply_t<1, false, true>(s, gvec, gv_out)
                       : (nt ? launch_apply_t<2, true, true>(s, gvec, gv_out) : launch_apply_t<2, false, true>(s, gvec, gv_out));
        else
            rc = !even ? launch_apply_t<1, false, false>(s, nullptr, nullptr)
                       : (nt ? launch_apply_t<2, true, false>(s, nullptr, nullptr)
                             : launch_apply_t<2, false, false>(s, nullptr, nullptr));
        if (rc) return rc;
        HIPCHK(hipGetLastError());
        s->dir ^= 1;
    }
    hipLaunchKernelGGL(k_pend_reset, dim3(64), dim3(256), 0, s->stream, s->d_pend, s->d_cpend,
                       (long long)(s->npend > s->defer ? MAXPEND : s->defer) * s->n, s->d_st);
    HIPCHK(hipGetLastError());
    s->npend = 0;
    return 0;
}

int launch_mirror_if_needed(ellhip_space* s) {
    if (!s->needs_mirror) return 0;
    const unsigned t = (unsigned)((s->n + 31) / 32);
    hipLaunchKernelGGL(k_mirror_lower, dim3(t, t), dim3(256), 0, s->stream, s->d_Q, s->ld, s->n, s->d_st);
    HIPCHK(hipGetLastError());
    return 0;
}

// EllStable, mirrored layout (ELLHIP_OPT_STABLE_SOLVE = 3; ellstable_kernels.hpp, StPend): entering copies the factor below
// the diagonal and sets every row scale to 1 (one 8 n^2-byte pass); leaving -- before anything observes the buffer (get_mq,
// clone), on a mode switch, when a halted queue's results are read, and every ST_MIRROR_PERIOD updates so that the running
// scales stay products of few factors -- rebuilds the scratch triangle of the last forward solve and scales U: the buffer is
// then what the eager kernels leave (to rounding: one rounding per element where they had two per update).
constexpr int ST_MIRROR_PERIOD = 256;
int stable_mirror_enter(ellhip_space* s) {
    const unsigned t = (unsigned)((s->n + 63) / 64);
    hipLaunchKernelGGL(k_st_mirror_enter, dim3(t, t), dim3(256), 0, s->stream, s->d_Q, s->ld, s->n, (const DevState*)s->d_st,
                       s->d_rbuf);
    hipLaunchKernelGGL(k_st_mirror_mark, dim3(1), dim3(1), 0, s->stream, s->d_stpend, (const DevState*)s->d_st);
    HIPCHK(hipGetLastError());
    s->st_mirrored = true;
    s->st_since_enter = 0;
    return 0;
}
int stable_mirror_leave(ellhip_space* s) {
    if (!s->st_mirrored) return 0;
    const long long n = s->n;
    const unsigned t = (unsigned)((n + 63) / 64);
    hipLaunchKernelGGL(k_st_mirror_leave, dim3(t, t), dim3(256), 0, s->stream, s->d_Q, s->ld, n, (const StPend*)s->d_stpend,
                       (const double*)s->d_rbuf, (const double*)s->d_wkeep);
    hipLaunchKernelGGL(k_st_unscale_upper, dim3((unsigned)std::min<long long>(16, (n + 255) / 256), (unsigned)n), dim3(256), 0,
                       s->stream, s->d_Q, s->ld, n, (const StPend*)s->d_stpend, (const double*)s->d_rbuf);
    hipLaunchKernelGGL(k_st_mirror_clear, dim3(1), dim3(1), 0, s->stream, s->d_stpend);
    HIPCHK(hipGetLastError());
    s->st_mirrored = false;
    return 0;
}

// EllStable::update_core as a fixed sequence of launches (ellstable_kernels.hpp).
// ---- one co-residency grid at a time per device ------------------------------------------------------------------------
// Two kinds of launches here consist of workgroups that WAIT for one another inside the launch and therefore assume all of
// them are on the device at once: a resident batch (k_ell_resident, one grid barrier per cut) and EllStable's persistent solves
// (chain / helper workgroups).  Two such grids at the same time -- two handles, two host threads -- can each hold part of the
// CUs and wait for workgroups of its own that the other keeps out: nothing hangs (every wait is bounded) but both time out.
// So within this process they are chained per device: whoever issues one waits, ON THE DEVICE, for the event the previous one
// left behind (hipStreamWaitEvent: no host synchronisation), and leaves its own.  Kernels that merely run long (the
// matrix-core passes draw tiles from a queue and keep their CUs for a whole pass) delay such a grid, they cannot lock it.
constexpr int CORES_MAX_DEVICES = 16;
constexpr int CORES_RING = 64;
struct CoresToken {
    std::mutex m;
    hipEvent_t ring[CORES_RING] = {};
    unsigned next = 0;
    hipEvent_t last = nullptr;
};
CoresToken g_cores[CORES_MAX_DEVICES];

struct CoresScope {  // from before the first such launch of an API call until after the last one has been enqueued
    CoresToken* t;
    hipStream_t stream;
    CoresScope(ellhip_space* s) : t(&g_cores[s->device & (CORES_MAX_DEVICES - 1)]), stream(s->stream) {
        t->m.lock();
        if (t->last) (void)hipStreamWaitEvent(stream, t->last, 0);
    }
    ~CoresScope() {
        hipEvent_t& e = t->ring[t->next % CORES_RING];
        if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
        if (e && hipEventRecord(e, stream) == hipSuccess) {
            t->last = e;
            t->next += 1;
        }
        t->m.unlock();
    }
    CoresScope(const CoresScope&) = delete;
    CoresScope& operator=(const CoresScope&) = delete;
};

int ellstable_issue(ellhip_space* s, const double* g_dev, const CutParams* cp_dev, CutParams cp_val, int queue_mode,
                    int* qst, double* qtsq) {
    const long long n = s->n, ld = s->ld;
    const long long nb = (n + SB - 1) / SB;
    double* w0 = s->d_work;
    double* z = w0 + n;
    double* gg = z + n;
    double* q = gg + n;
    double* beta2 = q + n;
    double* qpub = beta2 + n;            // publish buffer of the persistent backward solve (data-as-flag hand-off)
    double* w1 = qpub + n;               // second publish buffer of the persistent forward solve (see below)
    double* cpre = w1 + n + (n & 1);     // chunk prefixes of the mid stage (ST_MID_T + 1 doubles)
    hipStream_t st = s->stream;
    // One launch per solve when every workgroup of the chain can be resident at once (a workgroup that waits for its
    // predecessor holds its CU: the limit is what the DEVICE holds -- CU count x occupancy of the kernel, measured at
    // handle creation -- not a constant); otherwise one launch per block.
    const bool persist = s->stable_solve >= 1 && nb <= s->persist_cap1;
    // both solves with a helper workgroup per block (two workgroups per block, all resident): k_st_fwd_helped,
    // k_st_bwd_factor_helped (which also pulls the factor tiles)
    const bool helped = persist && s->stable_solve >= 2 && s->d_hpart && 2 * nb <= s->persist_cap_h;
    // Persistent forward solve: the workgroup that is next in the chain polls the VALUES of the block it waits for
    // (sentinel until stored, like the backward solve's qpub), everybody else the block's flag.  The published
    // vector therefore has to be all-sentinel when a solve starts: two buffers alternate by launch parity, and the
    // mid stage of every update re-arms the one the NEXT solve will use (whatever this update's status).
    // (one co-residency grid at a time on the device: chained behind the previous one, CoresScope)
    std::unique_ptr<CoresScope> alone;
    if (persist) alone.reset(new CoresScope(s));
    if (persist) ++s->epoch;
    double* w = (persist && (s->epoch & 1)) ? w1 : w0;
    double* w_next = (persist && (s->epoch & 1)) ? w0 : w1;
    int* err = reinterpret_cast<int*>(reinterpret_cast<char*>(s->d_st) + offsetof(DevState, solve_err));
    // The hand-over buffers of the in-launch waits are re-armed (all-sentinel) by the k_st_post of the update BEFORE the one
    // that polls them.  After a switch of forms (ellhip_set_option between two updates) that update may have run a form that
    // does not: its solve then left real values in w, and a consumer polling "the value is its own flag" would take the old
    // block for the new one.  Arm what this update polls and the previous one did not arm.
    {
        const unsigned ga = (unsigned)((n + 255) / 256);
        if (persist && !s->st_prev_persist) hipLaunchKernelGGL(k_st_arm, dim3(ga), dim3(256), 0, st, w, n);
        if (helped && !s->st_prev_helped) hipLaunchKernelGGL(k_st_arm, dim3(ga), dim3(256), 0, st, s->d_hpart, n);
        s->st_prev_persist = persist;
        s->st_prev_helped = helped;
    }
    // the helped backward solve (with or without factor tiles) needs its tile list / hand-over buffer and both grids resident
    const bool helped_b = helped && s->d_ftiles16 && s->d_fnext && s->d_qhpart && 2 * nb <= s->persist_cap1;
    // the mirrored layout (STABLE_SOLVE = 3): no scratch triangle, no factor pass -- both helped solves in their MIRROR form
    const bool mirror = helped_b && s->stable_solve >= 3 && s->d_stpend;
    // factor update inside the backward solve's launch: needs the helped form and the row-wise (U alone) arithmetic
    const bool fused_h = helped_b && !mirror && s->stable_factor >= 2;
    if (mirror && s->st_mirrored && s->st_since_enter >= ST_MIRROR_PERIOD) {
        int mrc = stable_mirror_leave(s);  // (and in again below: the row scales start from 1)
        if (mrc) return mrc;
    }
    if (mirror != s->st_mirrored) {
        int mrc = mirror ? stable_mirror_enter(s) : stable_mirror_leave(s);
        if (mrc) return mrc;
    }
    if (mirror) s->st_since_enter += 1;
    const StPend* pend_c = mirror ? s->d_stpend : nullptr;
    {
        ProfScope ps(s, CLS_ST_FWD);
        if (mirror) {
            hipLaunchKernelGGL(k_st_fwd_helped<true>, dim3((unsigned)(2 * nb)), dim3(256), 0, st, s->d_Q, ld, n, g_dev, w,
                               s->d_hpart, z, gg, s->d_flags, err, s->epoch, (const DevState*)s->d_st, pend_c,
                               (const double*)s->d_rbuf);
        } else if (helped) {
            hipLaunchKernelGGL(k_st_fwd_helped<false>, dim3((unsigned)(2 * nb)), dim3(256), 0, st, s->d_Q, ld, n, g_dev, w,
                               s->d_hpart, z, gg, s->d_flags, err, s->epoch, (const DevState*)s->d_st, (const StPend*)nullptr,
                               (const double*)nullptr);
        } else if (persist) {
            hipLaunchKernelGGL(k_st_fwd_persist, dim3((unsigned)nb), dim3(256), 0, st, s->d_Q, ld, n, g_dev, w, z, gg,
                               s->d_flags, err, s->epoch, s->d_st);
        } else {
            HIPCHK(hipMemcpyAsync(w, g_dev, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
            hipLaunchKernelGGL(k_st_fwd_first, dim3(1), dim3(256), 0, st, s->d_Q, ld, n, g_dev, w, z, gg, s->d_st);
            for (long long kb = 0; kb + 1 < nb; ++kb) {
                const long long rest = n - (kb + 1) * SB;
                const unsigned grid = (unsigned)((rest + SPANEL - 1) / SPANEL);
                hipLaunchKernelGGL(k_st_fwd_step, dim3(grid), dim3(256), 0, st, s->d_Q, ld, n, kb, w, z, gg, s->d_st);
            }
        }
        HIPCHK(hipGetLastError());
    }
    {
        ProfScope ps(s, CLS_SCALAR);
        EllCalcDev calc = EllCalcDev::make(n, s->use_parallel_cut);
        hipLaunchKernelGGL(k_st_mid, dim3(1), dim3(ST_MID_T), 0, st, n, (const double*)gg, cpre, s->d_st, calc, cp_dev,
                           cp_val, queue_mode, qst, qtsq, mirror ? s->d_stpend : (StPend*)nullptr);
        hipLaunchKernelGGL(k_st_post, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s->d_Q, ld, n,
                           (const double*)z, (const double*)gg, (const double*)cpre, q, beta2,
                           persist ? qpub : (double*)nullptr, persist ? w_next : (double*)nullptr,
                           (const DevState*)s->d_st, helped ? s->d_hpart : (double*)nullptr, s->d_fnext,
                           (fused_h || mirror) ? s->d_qhpart : (double*)nullptr, pend_c, s->d_rbuf, (const double*)w, s->d_wkeep);
        HIPCHK(hipGetLastError());
    }
    // The factor update (rewrites U) and the backward solve (reads S, writes q) are independent.  Helped form: the
    // factor tiles are pulled inside the backward solve's launch by whoever is idle.  Otherwise, with the persistent
    // backward solve (latency-bound) the bandwidth-bound factor update runs beside it on the auxiliary stream,
    // launched AFTER it so the solve's workgroups are placed first; the streams join before anything else touches Q.
    const bool overlap = persist && !fused_h && !mirror;
    if (overlap) HIPCHK(hipEventRecord(s->ev_fork, st));
    {
        ProfScope ps(s, CLS_ST_BWD);
        if (mirror) {
            hipLaunchKernelGGL((k_st_bwd_factor_helped<2048, 8, true>), dim3((unsigned)(2 * nb)), dim3(256), 0, st, s->d_Q, ld, n, q,
                               qpub, s->d_qhpart, err, (const DevState*)s->d_st, nb, (const double*)beta2, (const double*)w,
                               (const int*)s->d_ftiles16, s->nftiles16, s->d_fnext, nb + 1, pend_c, (const double*)s->d_rbuf);
        } else if (fused_h) {
            const unsigned grid = (unsigned)(2 * nb);
            // before their turn the chain workgroups pull factor tiles only when the matrix is on-die (n < 8192): from HBM
            // it made them late for their own block (n = 16384: stop distance 6 / 12 / 24 / 48+ blocks: 785 / 731 / 710 /
            // 685-695 us), on-die it pays (n = 4096: 139 us against 170 without)
            const long long fq_stop = n >= 8192 ? nb + 1 : FQ_STOP;
            if (n >= 8192)
                hipLaunchKernelGGL((k_st_bwd_factor_helped<2048, 8>), dim3(grid), dim3(256), 0, st, s->d_Q, ld, n, q, qpub,
                                   s->d_qhpart, err, (const DevState*)s->d_st, nb, (const double*)beta2, (const double*)w,
                                   (const int*)s->d_ftiles16, s->nftiles16, s->d_fnext, fq_stop);
            else
                hipLaunchKernelGGL((k_st_bwd_factor_helped<512, 8>), dim3(grid), dim3(256), 0, st, s->d_Q, ld, n, q, qpub,
                                   s->d_qhpart, err, (const DevState*)s->d_st, nb, (const double*)beta2, (const double*)w,
                                   (const int*)s->d_ftiles16, s->nftiles16, s->d_fnext, fq_stop);
        } else if (persist) {
            hipLaunchKernelGGL(k_st_bwd_persist, dim3((unsigned)nb), dim3(256), 0, st, s->d_Q, ld, n, q, qpub, err,
                               s->d_st);
        } else {
            hipLaunchKernelGGL(k_st_bwd_last, dim3(1), dim3(256), 0, st, s->d_Q, ld, n, nb - 1, q, s->d_st);
            for (long long kb = nb - 1; kb >= 1; --kb) {
                const unsigned grid = (unsigned)((kb * SB + SPANEL - 1) / SPANEL);
                hipLaunchKernelGGL(k_st_bwd_step, dim3(grid), dim3(256), 0, st, s->d_Q, ld, n, kb, q, s->d_st);
            }
        }
        const unsigned gx = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(k_st_xc, dim3(gx < 256 ? gx : 256), dim3(256), 0, st, n, q, s->d_xc, s->d_st);
        HIPCHK(hipGetLastError());
    }
    if (!fused_h && !mirror) {
        hipStream_t fs = overlap ? s->aux_stream : st;
        if (overlap) HIPCHK(hipStreamWaitEvent(fs, s->ev_fork, 0));
        {
            ProfScope ps(s, CLS_ST_FACTOR, fs);
            if (s->stable_factor >= 1) {
                // row-wise, from U alone (the scratch entry it would read IS fl(U * w): see k_st_factor_rows)
                const unsigned gy = (unsigned)((n + FROW_H - 1) / FROW_H);
                if (n >= 8192)
                    hipLaunchKernelGGL((k_st_factor_rows<2048, 2>), dim3(gy, (unsigned)((n + 2047) / 2048)), dim3(256), 0,
                                       fs, s->d_Q, ld, n, beta2, (const double*)w, s->d_st);
                else
                    hipLaunchKernelGGL((k_st_factor_rows<512, 4>), dim3(gy, (unsigned)((n + 511) / 512)), dim3(256), 0,
                                       fs, s->d_Q, ld, n, beta2, (const double*)w, s->d_st);
            } else {
                const unsigned nt64 = (unsigned)((n + 63) / 64);  // 64x64 tiles, scratch triangle transposed through LDS
                hipLaunchKernelGGL(k_st_factor, dim3(nt64, nt64), dim3(256), 0, fs, s->d_Q, ld, n, beta2, s->d_st);
            }
            HIPCHK(hipGetLastError());
        }
        if (overlap) {
            HIPCHK(hipEventRecord(s->ev_join, fs));
            HIPCHK(hipStreamWaitEvent(st, s->ev_join, 0));
        }
    }
    return 0;
}

// ---- the three primitives --------------------------------------------------------------------

// Lower-triangle GEMV: deferred mode (Q_base is bit-symmetric and only read).  Unsharded handles from
// symv_min_n up; a row shard only in its symmetric mode (ellhip_set_shard_symmetric), where the GEMV yields this
// shard's PARTIAL sums over its lower trapezoid and the caller adds the shards' vectors (all-reduce).
bool symv_ok(const ellhip_space* s) {
    if (!(deferring(s) && s->symv && (s->n % 2) == 0 && s->d_rowpart)) return false;
    return s->sharded ? s->shard_symmetric : s->n >= s->symv_min_n;
}

int symv_alloc(ellhip_space* s) {
    if (s->d_rowpart || (s->n % 2) != 0 || s->n < 512 || (s->sharded && !s->shard_symmetric)) return 0;
    // tiles of 64 x 2048 in this handle's lower trapezoid; with fewer than ~200 (less than one per CU) the narrow
    // segments win (measured per rank at n = 16384: P = 8, 128 tiles: 0.049 vs 0.061 ms; P = 4, 256 tiles: 0.078 vs
    // 0.067; P = 2: 0.109 vs 0.104; unsharded n = 8192, 256 tiles: 0.076 vs 0.066)
    const double area = ((double)(s->row0 + s->nrows) * (double)(s->row0 + s->nrows) - (double)s->row0 * (double)s->row0) / 2.0;
    s->symv_seg = (area / (64.0 * SYMV_SEG) < 200.0) ? SYMV_SEG_SMALL : SYMV_SEG;
    // rows in flight per thread: with about one 64 x 2048 tile per CU (n = 8192: 256 tiles) the workgroup itself has to
    // keep more loads in the air -- 4 rows: 61 vs 65 us per pass; with several tiles per CU (n = 16384) 2 and 4 tie
    s->symv_rw = (area / (64.0 * SYMV_SEG) < 600.0) ? 4 : 2;
    const size_t nsegs = (size_t)((s->n + s->symv_seg - 1) / s->symv_seg), nstrips = (size_t)((s->nrows + SYMV_H - 1) / SYMV_H);
    HIPCHK(hipMalloc(&s->d_rowpart, nsegs * (size_t)s->n * sizeof(double)));
    HIPCHK(hipMalloc(&s->d_colpart, nstrips * (size_t)s->n * sizeof(double)));
    // rows of rowpart outside this shard are never written but are read by nobody either; zero them anyway
    HIPCHK(hipMemsetAsync(s->d_rowpart, 0, nsegs * (size_t)s->n * sizeof(double), s->stream));
    HIPCHK(hipMemsetAsync(s->d_colpart, 0, nstrips * (size_t)s->n * sizeof(double), s->stream));
    return 0;
}

// partial-sum sets of the lower-triangle GEMV: 0 = the handle's own, 1 = the second set of overlapped queue runs,
// 2 + l = vector l of a k_symv_multi group
constexpr int MULTI_MAX = 32;  // = GRP_MAX = 2 * SMM_NV: gradients per matrix-core pass at most
constexpr int MULTI_VALU_MAX = 3;  // largest group k_symv_multi takes (vector ALU, bit-identical to k_symv)
size_t rowpart_elems(const ellhip_space* s) { return (size_t)((s->n + s->symv_seg - 1) / s->symv_seg) * (size_t)s->n; }
size_t colpart_elems(const ellhip_space* s) { return (size_t)((s->nrows + SYMV_H - 1) / SYMV_H) * (size_t)s->n; }
double* rowpart_of(const ellhip_space* s, int set) {
    return set == 0 ? s->d_rowpart : set == 1 ? s->d_rowpart2 : s->d_rowpart_m + (size_t)(set - 2) * rowpart_elems(s);
}
double* colpart_of(const ellhip_space* s, int set) {
    return set == 0 ? s->d_colpart : set == 1 ? s->d_colpart2 : s->d_colpart_m + (size_t)(set - 2) * colpart_elems(s);
}

template <int RW, int SEG>
void symv_go(ellhip_space* s, const double* g_dev, unsigned nstrips, unsigned nsegs, bool nt, hipStream_t st, int set) {
    double* rowpart = rowpart_of(s, set);
    double* colpart = colpart_of(s, set);
    if (nt)
        hipLaunchKernelGGL((k_symv<RW, true, 0, SEG>), dim3(nstrips, nsegs), dim3(256), 0, st, (const double*)s->d_Q,
                           s->ld, s->n, s->row0, s->nrows, g_dev, rowpart, colpart, s->d_st);
    else
        hipLaunchKernelGGL((k_symv<RW, false, 0, SEG>), dim3(nstrips, nsegs), dim3(256), 0, st, (const double*)s->d_Q,
                           s->ld, s->n, s->row0, s->nrows, g_dev, rowpart, colpart, s->d_st);
}

int launch_symv_reduce(ellhip_space* s, const double* g_dev, double* y_out);
// the tiles alone, on `st`, into partial-sum set `set`
int launch_symv_tiles(ellhip_space* s, const double* g_dev, hipStream_t st, int set) {
    const int seg = s->symv_seg;
    {
        ProfScope ps(s, CLS_SYMV, st);
        const unsigned nstrips = (unsigned)((s->nrows + SYMV_H - 1) / SYMV_H);  // local strips
        const unsigned nsegs = (unsigned)((s->n + seg - 1) / seg);
        const bool nt = s->sh_gemv.nt != 0;
        if (seg == SYMV_SEG_SMALL) {
            symv_go<8, SYMV_SEG_SMALL>(s, g_dev, nstrips, nsegs, nt, st, set);  // narrow segments: 8 rows x 1 chunk in flight
        } else if (seg == SYMV_SEG) {
            switch (s->symv_rw) {
                case 1: symv_go<1, SYMV_SEG>(s, g_dev, nstrips, nsegs, nt, st, set); break;
                case 2: symv_go<2, SYMV_SEG>(s, g_dev, nstrips, nsegs, nt, st, set); break;
                case 4: symv_go<4, SYMV_SEG>(s, g_dev, nstrips, nsegs, nt, st, set); break;
                case 8: symv_go<8, SYMV_SEG>(s, g_dev, nstrips, nsegs, nt, st, set); break;
                default: return fail(ELLHIP_E_INVALID, "unsupported rows-in-flight of the lower-triangle GEMV (1, 2, 4, 8)");
            }
        } else {
            return fail(ELLHIP_E_INVALID, "unsupported segment width of the lower-triangle GEMV (512, 2048)");
        }
        HIPCHK(hipGetLastError());
    }
    return 0;
}
int launch_symv(ellhip_space* s, const double* g_dev, double* y_out) {
    int rc = launch_symv_tiles(s, g_dev, s->stream, s->part_set);
    if (rc) return rc;
    return launch_symv_reduce(s, g_dev, y_out);
}

// the reduction of the tiles launch_symv has issued (by itself: k_symv_reduce<NP>)
int launch_symv_reduce(ellhip_space* s, const double* g_dev, double* y_out) {
    const int seg = s->symv_seg;
    ProfScope ps(s, CLS_SYMV_REDUCE);
    // unsharded: the reduction also yields the scalar stage's dot products (a shard's y is partial until the owner's
    // all-reduce has run, so its dot products wait for k_scalar_dot_def)
    const int np = (!s->sharded && s->fuse_dots) ? s->defer : 0;
#define REDUCE_GO(NPV)                                                                                                \
    hipLaunchKernelGGL(k_symv_reduce<NPV>, dim3((unsigned)((s->n + 127) / 128)), dim3(256), 0, s->stream, s->n, s->row0, \
                       s->nrows, (long long)seg, (const double*)rowpart_of(s, s->part_set),                          \
                       (const double*)colpart_of(s, s->part_set), y_out,                                               \
                       s->d_st, g_dev, (const double*)s->d_pend, s->d_partial)
    if (np == 24) REDUCE_GO(24);
    else if (np == 16) REDUCE_GO(16);
    else if (np == 8) REDUCE_GO(8);
    else REDUCE_GO(0);
#undef REDUCE_GO
    HIPCHK(hipGetLastError());
    s->dots_np = np;
    return 0;
}

// gt[slot] = Q * g  (Ell only; EllStable has no separate first pass)
int do_prime(ellhip_space* s, const double* g_dev, int slot) {
    if (s->variant != ELLHIP_SPACE_ELL) return 0;
    s->dots_np = 0;  // whatever d_partial held belonged to an earlier gradient
    s->dots_need_gy = false;
    if (symv_ok(s)) return launch_symv(s, g_dev, s->d_gt[slot]);
    if (s->shard_symmetric)
        return fail(ELLHIP_E_STATE, "symmetric row shard: only the deferred (depth 8) schedule is available");
    ProfScope ps(s, CLS_GEMV);
    // deferred depth 8 on full rows (no lower-triangle schedule at this size): the dot products ride along.  Up to
    // n = 8192 only: every workgroup of the scalar stage re-forms g.y from all of g and y, which stops paying beyond.
    if (deferring(s) && s->defer == 8 && !s->sharded && s->fuse_dots && s->n <= 8192)
        return launch_gemv_dots(s, g_dev, s->d_gt[slot]);
    return launch_sweep<false, true>(s, s->sh_gemv, nullptr, g_dev, s->d_gt[slot]);
}

// scalar stage of the primed cut (asynchronous; the caller reads the state back if it needs it)
int do_cut(ellhip_space* s, const double* g_dev, const CutParams* cp_dev, CutParams cp_val, int queue_mode, int* qst,
           double* qtsq) {
    s->xc_host_valid = false;  // the scalar stage moves the centre
    if (s->variant != ELLHIP_SPACE_ELL) return ellstable_issue(s, g_dev, cp_dev, cp_val, queue_mode, qst, qtsq);
    // The dot products a prime left in d_partial belong to THIS cut only: whatever path the cut takes (also the
    // non-deferred one, after a depth switch between prime and cut) they are spent now, and a gradient primed later
    // by a fused pass (rank-1 + GEMV, apply + GEMV) has none.
    const int dots_np_now = s->dots_np;
    s->dots_np = 0;
    ProfScope ps(s, CLS_SCALAR);
    EllCalcDev calc = EllCalcDev::make(s->n, s->use_parallel_cut);
    const unsigned G = (unsigned)scalar_groups(s->n);
    const double* gt = s->d_gt[s->cur];
    if (deferring(s)) {
        // gt currently holds y = Q_base * g; the stage corrects it with the pending updates and records the
        // cut as pending update number s->npend (optimistically counted; see ellhip_queue_results / callers)
        // The dot products come from k_symv_reduce when THIS gradient was primed by it at the depth in force
        // (dots_np); any other history (full-row GEMV, a shard, a depth or mode switch since) takes the separate launch.
        const bool have_dots = dots_np_now == s->defer;
        const bool gy_inside = have_dots && s->dots_need_gy;
        const int npart = (have_dots && !gy_inside) ? (int)((s->n + 127) / 128) : (int)G;
        // a live update: the stage's own workgroups push the centre and the state to the host (live_tail) -- no k_publish launch
        LiveMirror* lm = nullptr;
        unsigned long long lseq = 0;
        if (s->live_arm && queue_mode == 0) {
            s->live_seq += 1;
            lm = s->h_live;
            lseq = s->live_seq;
            s->live_done = true;
        }
#define SCALAR_DEF(NPV)                                                                                           \
    if (!have_dots)                                                                                               \
        hipLaunchKernelGGL(k_scalar_dot_def<NPV>, dim3(G), dim3(256), 0, s->stream, s->n, g_dev, gt,               \
                           (const double*)s->d_pend, s->d_partial, s->d_st);                                      \
    if (gy_inside)                                                                                                \
        hipLaunchKernelGGL((k_scalar_apply_def<NPV, true>), dim3(G), dim3(256), 0, s->stream, s->n, gt, s->d_xc,   \
                           s->d_pend, s->d_cpend, (const double*)s->d_partial, s->d_st, calc, cp_dev, cp_val,     \
                           s->npend, queue_mode, qst, qtsq, npart, g_dev, lm, s->h_xc, lseq, s->d_pub_arrived);   \
    else                                                                                                          \
        hipLaunchKernelGGL(k_scalar_apply_def<NPV>, dim3(G), dim3(256), 0, s->stream, s->n, gt, s->d_xc, s->d_pend, \
                           s->d_cpend, (const double*)s->d_partial, s->d_st, calc, cp_dev, cp_val, s->npend,      \
                           queue_mode, qst, qtsq, npart, (const double*)nullptr, lm, s->h_xc, lseq, s->d_pub_arrived)
        if (s->defer == 24) { SCALAR_DEF(24); } else if (s->defer == 16) { SCALAR_DEF(16); } else { SCALAR_DEF(8); }
#undef SCALAR_DEF
        HIPCHK(hipGetLastError());
        s->npend += 1;
        return 0;
    }
    if (s->n < SCALAR_SPLIT_N) {
        hipLaunchKernelGGL(k_scalar, dim3(1), dim3(1024), 0, s->stream, s->n, g_dev, gt, s->d_xc, s->d_st, calc,
                           cp_dev, cp_val, s->no_defer_trick, queue_mode, qst, qtsq);
        HIPCHK(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(k_scalar_dot, dim3(G), dim3(256), 0, s->stream, s->n, g_dev, gt, s->d_partial, s->d_st);
    hipLaunchKernelGGL(k_scalar_apply, dim3(G), dim3(256), 0, s->stream, s->n, gt, s->d_xc,
                       (const double*)s->d_partial, s->d_st, calc, cp_dev, cp_val, s->no_defer_trick, queue_mode,
                       qst, qtsq);
    HIPCHK(hipGetLastError());
    return 0;
}

// rank-1 pass for the cut just taken (the kernel itself skips it when the cut failed), fused with
// the GEMV pass of `gnext_dev` (into the other slot) when that is given.
int do_commit(ellhip_space* s, bool shrink, const double* gnext_dev) {
    if (s->variant != ELLHIP_SPACE_ELL) return 0;
    if (deferring(s)) {
        // nothing to shrink now: the cut was recorded.  When the slots are full, one pass applies them all
        // (and carries the next GEMV); otherwise the next gradient only needs a read-only pass.
        if (s->npend >= s->defer) {
            if (gnext_dev && !symv_ok(s)) return flush_pending(s, gnext_dev, s->d_gt[s->cur ^ 1]);
            int rc = flush_pending(s, nullptr, nullptr);  // the next GEMV runs on the lower triangle afterwards
            if (rc) return rc;
        }
        if (gnext_dev) return do_prime(s, gnext_dev, s->cur ^ 1);
        return 0;
    }
    if (shrink) {
        int rc = launch_mirror_if_needed(s);
        if (rc) return rc;
    }
    const double* gt_cur = s->d_gt[s->cur];
    if (shrink && gnext_dev) {
        ProfScope ps(s, CLS_FUSED);
        return launch_sweep<true, true>(s, s->sh_fused, gt_cur, gnext_dev, s->d_gt[s->cur ^ 1]);
    }
    if (shrink) {
        ProfScope ps(s, CLS_RANK1);
        return launch_sweep<true, false>(s, s->sh_rank1, gt_cur, nullptr, nullptr);
    }
    if (gnext_dev) return do_prime(s, gnext_dev, s->cur ^ 1);
    return 0;
}

int read_back(ellhip_space* s) {
    HIPCHK(hipMemcpyAsync(s->h_result, s->d_st, sizeof(DevState), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    s->kappa = s->h_result->kappa;
    s->tsq = s->h_result->tsq;
    s->scalars_stale = false;
    if (s->h_result->solve_err) {
        // A bounded wait of a persistent solve gave up: this update's result is invalid and the call fails.  The
        // error word is cleared once it has been reported and the handle falls back to one launch per block (no
        // inter-workgroup waits), so the handle stays usable -- its state, though, is what the failed update left.
        // (Ell: resident batches settle their own time-outs inside resident_run and leave no error word behind.)
        s->h_result->solve_err = 0;
        (void)hipMemsetAsync(reinterpret_cast<char*>(s->d_st) + offsetof(DevState, solve_err), 0, sizeof(int), s->stream);
        (void)hipStreamSynchronize(s->stream);
        if (s->variant == ELLHIP_SPACE_ELL) {
            s->resident = 0;
            return fail(ELLHIP_E_HIP, "a bounded in-launch wait timed out on an Ell handle; resident batches are now off on it");
        }
        s->stable_solve = 0;
        (void)stable_mirror_leave(s);  // (flags only matter from here on: the buffer is what the failed update left)
        return fail(ELLHIP_E_HIP, "a bounded in-launch wait of an EllStable persistent solve timed out; this handle now uses one "
                                  "launch per block (no inter-workgroup waits)");
    }
    return 0;
}

// ---- live updates: what the caller's next statement needs, and no more ---------------------------------------------
// live_publish is enqueued right behind the scalar stage of a direct update; live_wait returns as soon as that update's
// status / tsq / kappa and the new centre are in host memory.  Whatever the update still has in flight behind it (the
// rank-1 pass, an apply pass of the recorded schedule) keeps running: the next call on the handle is ordered after it by
// the stream.  Same observable contract as read_back (h_result, kappa, tsq, the solve_err protocol).
int live_publish(ellhip_space* s) {
    if (s->live_done) {   // the scalar stage has published itself (live_tail)
        s->live_done = false;
        return 0;
    }
    s->live_seq += 1;
    const unsigned wgs = (unsigned)std::max<long long>(1, std::min<long long>(PUB_WGS, s->n / 1024));
    hipLaunchKernelGGL(k_publish, dim3(wgs), dim3(256), 0, s->stream, (const DevState*)s->d_st, (const double*)s->d_xc, s->n,
                       s->h_live, s->h_xc, s->live_seq, s->d_pub_arrived);
    HIPCHK(hipGetLastError());
    return 0;
}
int live_wait(ellhip_space* s) {
    const unsigned long long want = s->live_seq;
    const unsigned long long* seq = &s->h_live->seq;
    for (unsigned spin = 1;; ++spin) {
        if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) == want) break;
        if ((spin & 0x3fffu) == 0) {  // now and then: is the stream still alive?
            const hipError_t q = hipStreamQuery(s->stream);
            if (q == hipSuccess) {    // drained: the word must be there
                if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) == want) break;
                return fail(ELLHIP_E_HIP, "live update: the stream drained without publishing its result");
            }
            if (q != hipErrorNotReady) return fail(ELLHIP_E_HIP, "live update", q);
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    *s->h_result = s->h_live->st;
    s->kappa = s->h_result->kappa;
    s->tsq = s->h_result->tsq;
    s->scalars_stale = false;
    s->xc_host_valid = true;
    if (s->h_result->solve_err) return read_back(s);  // (rare: the full protocol, with the stream drained)
    return 0;
}

// upload a host gradient into stage slot `slot`
int stage_grad(ellhip_space* s, const double* grad, int slot) {
    if (!grad) return fail(ELLHIP_E_INVALID, "grad is NULL");
    const size_t bytes = (size_t)s->n * sizeof(double);
    if (s->stage_direct) {
        // Nothing in flight reads this slot: every reader of a staged gradient (the GEMV, the scalar stage, EllStable's forward
        // solve) has finished before the call that consumed it returned, and the other slot is the one a primed gradient holds.
        memcpy(s->d_stage[slot], grad, bytes);
        __atomic_thread_fence(__ATOMIC_SEQ_CST);   // (write-combined stores drained before the launch's doorbell)
        return 0;
    }
    memcpy(s->h_stage[slot], grad, bytes);
    // (the chip pulls the staging buffer over PCIe itself: no copy engine, no cross-engine hand-over in front of the GEMV)
    const unsigned wgs = (unsigned)std::max<long long>(1, std::min<long long>(STAGE_WGS, s->n / 512));
    hipLaunchKernelGGL(k_stage, dim3(wgs), dim3(256), 0, s->stream, (const double*)s->h_stage[slot], s->d_stage[slot], s->n);
    HIPCHK(hipGetLastError());
    return 0;
}

int make_params(int kind, double b0, int has_b1, double b1, CutParams& cp) {
    if (kind < 0 || kind > 2) return fail(ELLHIP_E_INVALID, "bad cut kind");
    cp.kind = kind;
    cp.has_b1 = has_b1 ? 1 : 0;
    cp.b0 = b0;
    cp.b1 = has_b1 ? b1 : 0.0;
    return 0;
}

// Make Q current: issue a shrink that a previous ellhip_cut / queue cut left pending.
int ensure_committed(ellhip_space* s) {
    if (s->in_two_phase) return fail(ELLHIP_E_STATE, "update_begin without update_end");
    if (s->shrink_pending) {
        int rc = do_commit(s, true, nullptr);
        if (rc) return rc;
        s->shrink_pending = false;
    }
    return 0;
}

void drop_prime(ellhip_space* s);

// A gradient that is primed but not yet cut carries y = Q_base * g for the base the recorded updates belong to
// (the scalar stage corrects it with them).  When something other than the cut sequence itself applies the
// recorded updates -- ellhip_flush, or an observer of Q between commit(next) and cut(next) -- the base changes
// under that y, so it is recomputed against the new base.  A row shard cannot (its GEMV needs the owner's
// collective): its prime is dropped instead and the owner primes again (ellhip_queue_primed tells).
bool prime_is_uncut(const ellhip_space* s) {
    return s->variant == ELLHIP_SPACE_ELL && s->primed && !s->shrink_pending && s->g_cur != nullptr;
}
int refresh_prime(ellhip_space* s) {
    if (s->sharded) {
        drop_prime(s);
        return 0;
    }
    return do_prime(s, s->g_cur, s->cur);
}

// For observers of Q itself (get_mq, clone, mode switches): also apply what deferred mode has recorded.
int make_q_current(ellhip_space* s) {
    if (s->variant == ELLHIP_SPACE_ELL_STABLE) {  // (the buffer in the reference's layout)
        const int erc = ensure_committed(s);
        return erc ? erc : stable_mirror_leave(s);
    }
    const bool uncut = prime_is_uncut(s);
    int rc = ensure_committed(s);
    if (rc) return rc;
    if (s->npend > 0) {
        rc = flush_pending(s, nullptr, nullptr);
        if (rc) return rc;
        if (uncut) {
            rc = refresh_prime(s);
            if (rc) return rc;
        }
    }
    if (s->upper_stale && !s->sharded) {  // lower-triangle-only apply passes ran: rebuild the mirrored half
        // (a symmetric row shard cannot: the mirrored elements live on other ranks; its rows stay valid up to
        // their diagonal only, see ellhip_set_shard_symmetric)
        const unsigned t = (unsigned)((s->n + 31) / 32);
        hipLaunchKernelGGL(k_mirror_lower_now, dim3(t, t), dim3(256), 0, s->stream, s->d_Q, s->ld, s->n);
        HIPCHK(hipGetLastError());
        s->upper_stale = false;
    }
    return 0;
}

void drop_prime(ellhip_space* s) {
    s->primed = false;
    s->g_cur = nullptr;
    s->primed_qindex = -1;
}

int alloc_common(ellhip_space* s) {
    const long long n = s->n;
    const size_t vbytes = (size_t)n * sizeof(double);
    HIPCHK(hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking));
    s->stream = s->own_stream;
    HIPCHK(hipMalloc(&s->d_Q, (size_t)s->nrows * (size_t)s->ld * sizeof(double)));
    HIPCHK(hipMalloc(&s->d_xc, vbytes));
    {   // Large-BAR systems: the host writes a gradient straight into device memory (fine-grained allocation: 3 us for 128 KiB)
        // instead of into a pinned buffer that a kernel then pulls over PCIe (0.8 + 7.6 us and a launch on the live loop's
        // critical path, tools/experiments/bar_write.hip).
        int large_bar = 0;
        if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, s->device) != hipSuccess) large_bar = 0;
        (void)hipGetLastError();
        s->stage_direct = large_bar != 0 && g_defaults.stage_direct != 0;
    }
    if (s->stage_direct) {
        for (int k = 0; k < 2 && s->stage_direct; ++k)
            if (hipExtMallocWithFlags(reinterpret_cast<void**>(&s->d_stage[k]), vbytes, hipDeviceMallocFinegrained) != hipSuccess) {
                (void)hipGetLastError();
                s->d_stage[k] = nullptr;
                s->stage_direct = false;
            }
        if (!s->stage_direct)
            for (int k = 0; k < 2; ++k) {
                if (s->d_stage[k]) (void)hipFree(s->d_stage[k]);
                s->d_stage[k] = nullptr;
            }
    }
    if (!s->stage_direct)
        for (int k = 0; k < 2; ++k) HIPCHK(hipMalloc(&s->d_stage[k], vbytes));
    for (int k = 0; k < 2; ++k) {
        HIPCHK(hipMalloc(&s->d_gt_own[k], vbytes));
        s->d_gt[k] = s->d_gt_own[k];
        HIPCHK(hipHostMalloc(&s->h_stage[k], vbytes, hipHostMallocCoherent | hipHostMallocMapped));  // (k_stage reads it from the device)
        HIPCHK(hipMemsetAsync(s->d_gt_own[k], 0, vbytes, s->stream));
    }
    HIPCHK(hipMalloc(&s->d_st, sizeof(DevState)));
    // partial sums of the scalar stage: [scalar_groups(n) <= 64][MAXPEND + 1], or [ceil(n / 128)][depth + 1] when the
    // lower-triangle GEMV's reduction produces them
    HIPCHK(hipMalloc(&s->d_partial, (size_t)std::max<long long>(64, (n + 127) / 128) * (MAXPEND + 1) * sizeof(double)));
    if (s->variant == ELLHIP_SPACE_ELL) {
        HIPCHK(hipMalloc(&s->d_pend, (size_t)MAXPEND * vbytes));
        HIPCHK(hipMalloc(&s->d_cpend, MAXPEND * sizeof(double)));
        HIPCHK(hipMemsetAsync(s->d_pend, 0, (size_t)MAXPEND * vbytes, s->stream));
        HIPCHK(hipMemsetAsync(s->d_cpend, 0, MAXPEND * sizeof(double), s->stream));
    }
    if (s->variant == ELLHIP_SPACE_ELL_STABLE) {
        const size_t nflags = (size_t)std::max<long long>(512, (n + SB - 1) / SB);
        HIPCHK(hipMalloc(&s->d_flags, nflags * sizeof(int)));
        HIPCHK(hipMemsetAsync(s->d_flags, 0, nflags * sizeof(int), s->stream));
        {   // how many workgroups of each persistent solve this device keeps resident at once
            hipDeviceProp_t prop;
            HIPCHK(hipGetDeviceProperties(&prop, s->device));
            auto cap = [&](const void* fwd, const void* bwd, int threads) -> int {
                int a = 0, b = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, fwd, threads, 0) != hipSuccess) a = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, bwd, threads, 0) != hipSuccess) b = 0;
                return std::min(a, b) * prop.multiProcessorCount;
            };
            s->persist_cap1 = cap((const void*)k_st_fwd_persist, (const void*)k_st_bwd_persist, 256);
            s->persist_cap_h = std::min(cap((const void*)k_st_fwd_helped<false>, (const void*)k_st_bwd_factor_helped<2048, 8>, 256),
                                        cap((const void*)k_st_fwd_helped<true>, (const void*)k_st_bwd_factor_helped<2048, 8, true>, 256));
        }
        {   // the 16-row factor tiles of k_st_bwd_factor_helped: the active tiles of the strict upper triangle, full ones first
            const long long seg = (n >= 8192) ? 2048 : 512;
            const long long nseg = (n + seg - 1) / seg;
            std::vector<int> f16, e16;
            const long long ngrp = (n + FQ_H - 1) / FQ_H;
            for (long long I = 0; I < ngrp && nseg <= 16; ++I)
                for (long long J = 0; J < nseg; ++J) {
                    const long long r0 = I * FQ_H, c0 = J * seg;
                    if (c0 + seg - 1 <= r0) continue;
                    const long long rlast = std::min(r0 + FQ_H - 1, n - 1);
                    ((c0 > rlast && c0 + seg <= n) ? f16 : e16).push_back((int)((I << 4) | J));
                }
            f16.insert(f16.end(), e16.begin(), e16.end());
            s->nftiles16 = (int)f16.size();
            HIPCHK(hipMalloc(&s->d_fnext, sizeof(int)));
            HIPCHK(hipMemsetAsync(s->d_fnext, 0, sizeof(int), s->stream));
            if (s->nftiles16 > 0) {
                HIPCHK(hipMalloc(&s->d_ftiles16, f16.size() * sizeof(int)));
                HIPCHK(hipMemcpyAsync(s->d_ftiles16, f16.data(), f16.size() * sizeof(int), hipMemcpyHostToDevice, s->stream));
                HIPCHK(hipStreamSynchronize(s->stream));  // `f16` goes out of scope
                HIPCHK(hipMalloc(&s->d_qhpart, vbytes));
                hipLaunchKernelGGL(k_st_arm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, s->d_qhpart, n);
            }
        }
        HIPCHK(hipStreamCreateWithFlags(&s->aux_stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
    }
    if (s->variant == ELLHIP_SPACE_ELL_STABLE) {
        HIPCHK(hipMalloc(&s->d_work, vbytes * 8 + (ST_MID_T + 8) * sizeof(double)));  // w0 z gg q beta2 qpub w1 (+1 spare) | cpre
        // both publish buffers of the persistent forward solve start out all-sentinel
        hipLaunchKernelGGL(k_st_arm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, s->d_work, n);
        hipLaunchKernelGGL(k_st_arm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, s->d_work + 6 * n, n);
        HIPCHK(hipMalloc(&s->d_hpart, vbytes));
        hipLaunchKernelGGL(k_st_arm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, s->d_hpart, n);
        HIPCHK(hipMalloc(&s->d_stpend, sizeof(StPend)));
        HIPCHK(hipMalloc(&s->d_rbuf, 2 * vbytes));
        HIPCHK(hipMalloc(&s->d_wkeep, vbytes));
        hipLaunchKernelGGL(k_st_mirror_clear, dim3(1), dim3(1), 0, s->stream, s->d_stpend);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipHostMalloc(&s->h_result, sizeof(DevState), hipHostMallocDefault));
    HIPCHK(hipHostMalloc(&s->h_live, sizeof(LiveMirror), hipHostMallocCoherent | hipHostMallocMapped));
    HIPCHK(hipHostMalloc(&s->h_xc, vbytes, hipHostMallocCoherent | hipHostMallocMapped));
    memset(s->h_live, 0, sizeof(LiveMirror));
    HIPCHK(hipMalloc(&s->d_pub_arrived, sizeof(unsigned)));
    HIPCHK(hipMemsetAsync(s->d_pub_arrived, 0, sizeof(unsigned), s->stream));
    return 0;
}

int write_state(ellhip_space* s) {
    DevState st;
    memset(&st, 0, sizeof st);
    st.kappa = s->kappa;
    st.tsq = s->tsq;
    st.scale = 1.0;
    st.status = ELLHIP_SUCCESS;
    st.tol = -1.0;
    *s->h_result = st;
    HIPCHK(hipMemcpyAsync(s->d_st, s->h_result, sizeof(DevState), hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}

// Bitwise symmetry test of the caller's matrix, in 64 x 64 tiles so that the transposed accesses stay in cache
// (a plain double loop walks one operand with stride n: tens of seconds at n = 16384).
bool host_is_symmetric(const double* mq, long long n) {
    constexpr long long T = 64;
    for (long long i0 = 0; i0 < n; i0 += T)
        for (long long j0 = 0; j0 <= i0; j0 += T) {
            const long long i1 = std::min(i0 + T, n), j1 = std::min(j0 + T, n);
            for (long long i = i0; i < i1; ++i)
                for (long long j = j0; j < j1 && j < i; ++j)
                    if (memcmp(&mq[i * n + j], &mq[j * n + i], sizeof(double)) != 0) return false;
        }
    return true;
}

int create_impl(ellhip_space** out, int variant, long long n, long long row0, long long nrows, bool sharded,
                double kappa, const double* mq, const double* diag, const double* xc, int device) {
    if (!out) return fail(ELLHIP_E_INVALID, "out is NULL");
    *out = nullptr;
    if (n < 1 || nrows < 1 || row0 < 0 || row0 + nrows > n) return fail(ELLHIP_E_INVALID, "bad dimensions");
    if (variant != ELLHIP_SPACE_ELL && variant != ELLHIP_SPACE_ELL_STABLE)
        return fail(ELLHIP_E_INVALID, "unknown variant");
    if (sharded && variant != ELLHIP_SPACE_ELL)
        return fail(ELLHIP_E_INVALID, "EllStable does not shard (replicas only)");
    int ndev = ellhip_device_count();
    if (ndev <= 0) return fail(ELLHIP_E_NODEVICE, "no HIP device: the ellipsoid engine has no CPU path");
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= ndev) return fail(ELLHIP_E_INVALID, "device index out of range");

    ellhip_space* s = new (std::nothrow) ellhip_space();
    if (!s) return fail(ELLHIP_E_NOMEM, "host allocation failed");
    s->variant = variant;
    s->n = n;
    s->row0 = row0;
    s->nrows = nrows;
    s->sharded = sharded;
    s->device = device;
    s->kappa = kappa;
    s->tsq = 0.0;
    // Leading dimension: rows stay 16-byte aligned for even n.  For blocks that stream from HBM a
    // power-of-two row pitch is broken up with one extra 128-byte line per row (GEMV pass at
    // n = 16384: 5.8 -> 6.3 TB/s); blocks that live in the Infinity Cache are left dense.
    s->ld = n;
    if ((n % 512) == 0 && (double)nrows * (double)n * 8.0 > 200.0 * 1024 * 1024) s->ld = n + 16;
    if (g_defaults.pad >= 0) s->ld = n + g_defaults.pad;  // ELLHIP_OPT_PAD (tuning runs)
    if ((n % 2) == 0 && (s->ld % 2) != 0) s->ld += 1;
    if (variant == ELLHIP_SPACE_ELL_STABLE) s->ld = n + (n & 1);  // 16-byte aligned rows for the 2-column lanes
    pick_shape(s);

    DeviceGuard guard(device);
    int rc = alloc_common(s);
    if (rc) {
        ellhip_destroy(s);
        return rc;
    }
    auto bail = [&](int code) {
        ellhip_destroy(s);
        return code;
    };
    // centre
    if (xc) {
        memcpy(s->h_stage[0], xc, (size_t)n * sizeof(double));
        if (hipMemcpyAsync(s->d_xc, s->h_stage[0], (size_t)n * sizeof(double), hipMemcpyHostToDevice, s->stream) !=
            hipSuccess)
            return bail(fail(ELLHIP_E_HIP, "upload xc"));
        if (hipStreamSynchronize(s->stream) != hipSuccess) return bail(fail(ELLHIP_E_HIP, "sync"));
    } else {
        if (hipMemsetAsync(s->d_xc, 0, (size_t)n * sizeof(double), s->stream) != hipSuccess)
            return bail(fail(ELLHIP_E_HIP, "memset xc"));
    }
    // matrix
    if (mq) {
        if (hipMemcpy2DAsync(s->d_Q, (size_t)s->ld * sizeof(double), mq, (size_t)n * sizeof(double),
                             (size_t)n * sizeof(double), (size_t)nrows, hipMemcpyHostToDevice,
                             s->stream) != hipSuccess)
            return bail(fail(ELLHIP_E_HIP, "upload mq"));
        if (s->ld != n) {
            // zero the padding columns so they never hold NaNs
            if (hipMemset2DAsync(s->d_Q + n, (size_t)s->ld * sizeof(double), 0, (size_t)(s->ld - n) * sizeof(double),
                                 (size_t)nrows, s->stream) != hipSuccess)
                return bail(fail(ELLHIP_E_HIP, "memset pad"));
        }
        if (variant == ELLHIP_SPACE_ELL && !sharded) s->needs_mirror = !host_is_symmetric(mq, n);
    } else {
        double* d_diag = nullptr;
        if (diag) {
            memcpy(s->h_stage[0], diag, (size_t)n * sizeof(double));
            if (hipMemcpyAsync(s->d_stage[0], s->h_stage[0], (size_t)n * sizeof(double), hipMemcpyHostToDevice,
                               s->stream) != hipSuccess)
                return bail(fail(ELLHIP_E_HIP, "upload diag"));
            d_diag = s->d_stage[0];
        }
        hipLaunchKernelGGL(k_fill_diag, dim3(2048), dim3(256), 0, s->stream, s->d_Q, s->ld, n, nrows, row0,
                           (const double*)d_diag);
        if (hipGetLastError() != hipSuccess) return bail(fail(ELLHIP_E_HIP, "k_fill_diag"));
    }
    if (hipStreamSynchronize(s->stream) != hipSuccess) return bail(fail(ELLHIP_E_HIP, "sync after upload"));
    rc = write_state(s);
    if (rc) return bail(rc);
    // Default schedule of a new unsharded Ell handle: depth 24 wherever the lower-triangle schedule exists (even
    // n >= 5120 (ELLHIP_OPT_SYMV_MIN_N): 4.33 n^2 instead of 24 n^2 bytes per update, the recorded updates applied as one rank-24 update on the
    // matrix cores; results within the parity tolerance of depth 1, Q made current for every observer), otherwise the
    // reference's data flow (depth 1).  ELLHIP_OPT_AUTO_DEFER = 0 keeps depth 1 everywhere; ellhip_set_defer_depth overrides either way.  Row shards stay at 1 until their owner chooses.
    // Between 3072 and that size (and for odd n) depth 8 with full-row GEMVs is the faster one, for synchronous calls
    // and for queues alike (tools/depth_sweep.py: n = 4096: 15 800 vs 10 400 calls/s; below ~3000 depth 1 wins).
    if (variant == ELLHIP_SPACE_ELL && !sharded && g_defaults.auto_defer) {
        const bool lower = s->symv && s->apply_lower && (n % 2) == 0 && n >= s->symv_min_n;
        const int depth = lower ? 24 : (n >= 3072 ? 8 : 1);
        if (depth != 1) {
            rc = ellhip_set_defer_depth(s, depth);
            if (rc) return bail(rc);
        }
    }
    *out = s;
    return 0;
}

void queue_free(ellhip_space* s) {
    if (s->d_qparams) (void)hipFree(s->d_qparams);
    if (s->d_qgrads) (void)hipFree(s->d_qgrads);
    if (s->d_qstatus) (void)hipFree(s->d_qstatus);
    if (s->d_qtsq) (void)hipFree(s->d_qtsq);
    s->d_qparams = nullptr;
    s->d_qgrads = nullptr;
    s->d_qstatus = nullptr;
    s->d_qtsq = nullptr;
    s->qk = 0;
}

const double* qgrad(const ellhip_space* s, long long i) { return s->d_qgrads + (size_t)i * (size_t)s->n; }

int queue_index_ok(const ellhip_space* s, long long i) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    if (i < 0 || i >= s->qk) return fail(ELLHIP_E_INVALID, "queue index out of range");
    return 0;
}

// queue: make sure gt[cur] = Q * grad[index] (no-op when the previous commit already fused it)
int queue_prime_impl(ellhip_space* s, long long index) {
    if (s->primed && s->primed_qindex == index) return 0;
    int rc = ensure_committed(s);
    if (rc) return rc;
    rc = do_prime(s, qgrad(s, index), s->cur);
    if (rc) return rc;
    s->primed = true;
    s->g_cur = qgrad(s, index);
    s->primed_qindex = index;
    return 0;
}

int queue_cut_impl(ellhip_space* s, long long index) {
    if (!(s->primed && s->primed_qindex == index)) return fail(ELLHIP_E_STATE, "queue_cut: this cut is not primed");
    CutParams none{};
    int rc = do_cut(s, s->g_cur, s->d_qparams + index, none, 1, s->d_qstatus + index, s->d_qtsq + index);
    if (rc) return rc;
    s->shrink_pending = true;  // whether it really applies is decided on the device (DevState.apply)
    s->scalars_stale = true;   // kappa / tsq now live on the device until the next read-back
    return 0;
}

int queue_commit_impl(ellhip_space* s, long long index, long long next) {
    (void)index;
    const double* gnext = (next >= 0) ? qgrad(s, next) : nullptr;
    int rc = do_commit(s, s->shrink_pending, gnext);
    if (rc) return rc;
    s->shrink_pending = false;
    if (gnext) {
        s->cur ^= 1;
        s->primed = true;
        s->g_cur = gnext;
        s->primed_qindex = next;
    } else {
        drop_prime(s);
    }
    return 0;
}


// ---- pipelined queue runs with the next GEMV issued ahead (ELLHIP_OPT_OVERLAP) ---------------------------------------
// On the recorded schedule y = Q_base * g of the NEXT queued cut does not depend on the cut being taken (Q_base only
// changes in an apply pass; the scalar stage corrects y with the recorded updates afterwards), and the queue holds the
// next gradient already.  So k_symv of cut i+1 goes to a second, low-priority stream BEFORE the scalar stage of cut i is
// enqueued, into the other set of partial sums; its reduction (which also forms the dot products with the vector cut i
// records) follows on the main stream once both are done.  The 27 us of reduction + scalar stage per cut then run beside
// the 190 us GEMV instead of between two of them.  Same kernels, same operands, same order of every sum: bit-identical
// to the serial issue order.  Not across an apply pass (the GEMV after it reads the matrix it writes), not on shards
// (their collective sits between the GEMV and the scalar stage), not while profiling events would be recorded out of
// order -- ProfScope takes the stream a kernel is launched on.
bool overlap_ok(const ellhip_space* s) {
    return s->overlap && s->variant == ELLHIP_SPACE_ELL && !s->sharded && symv_ok(s);
}

// ---- the second stream: three functions carry every cross-stream ordering of the queue runs -------------------------
// queue_run_overlapped (the next cut's GEMV) and queue_run_multi (the next group's products) issue kernels that READ Q
// and WRITE one half of the partial-sum sets on a second stream beside the handle's own.  The rules:
//   side_fork   what is issued on the second stream from here on follows EVERYTHING enqueued on the handle's stream so
//               far -- whatever wrote Q (apply passes of this loop, of a cut taken by itself, of ensure_committed) and
//               whatever read the half about to be overwritten.  Not "the events that matter": both ordering bugs of
//               round 3 were a writer of Q that one of several hand-placed events did not cover.
//   side_mark   the side kernels have been issued (records their completion event).
//   side_join   the handle's stream waits for them.  Called before their results are read, by every writer of Q on the
//               handle's stream (flush_pending) and when a queue run returns (SideGuard): nothing outside a queue run
//               ever sees side work in flight.
// ELLHIP_OPT_OVERLAP = 2 issues the "side" kernels on the handle's own stream (same kernels, same order of issue, one
// stream): the serial form the option-mix walks compare the overlapped one against, seed by seed.
int side_setup(ellhip_space* s) {
    if (s->symv_stream) return 0;
    int least = 0, greatest = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIPCHK(hipEventCreateWithFlags(&s->ev_symv, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&s->ev_side_go, hipEventDisableTiming));
    // lowest priority: the short reduction / scalar kernels of the main stream get the CU slots the GEMV's workgroups free
    HIPCHK(hipStreamCreateWithPriority(&s->symv_stream, hipStreamNonBlocking, least));
    return 0;
}
hipStream_t side_stream(const ellhip_space* s) { return s->overlap == 2 ? s->stream : s->symv_stream; }
int side_join(ellhip_space* s) {
    if (!s->side_busy) return 0;
    s->side_busy = false;
    if (side_stream(s) != s->stream) HIPCHK(hipStreamWaitEvent(s->stream, s->ev_symv, 0));
    return 0;
}
int side_fork(ellhip_space* s) {
    int rc = side_join(s);  // (one batch of side work at a time)
    if (rc) return rc;
    if (side_stream(s) != s->stream) {
        HIPCHK(hipEventRecord(s->ev_side_go, s->stream));
        HIPCHK(hipStreamWaitEvent(s->symv_stream, s->ev_side_go, 0));
    }
    return 0;
}
int side_mark(ellhip_space* s) {
    if (side_stream(s) != s->stream) HIPCHK(hipEventRecord(s->ev_symv, side_stream(s)));
    s->side_busy = true;
    return 0;
}
struct SideGuard {  // a queue run never returns (error paths included) with side work in flight
    ellhip_space* s;
    ~SideGuard() { (void)side_join(s); }
};

int overlap_setup(ellhip_space* s) {
    if (s->d_rowpart2) return 0;
    int rc = side_setup(s);
    if (rc) return rc;
    const size_t nsegs = (size_t)((s->n + s->symv_seg - 1) / s->symv_seg), nstrips = (size_t)((s->nrows + SYMV_H - 1) / SYMV_H);
    HIPCHK(hipMalloc(&s->d_rowpart2, nsegs * (size_t)s->n * sizeof(double)));
    HIPCHK(hipMalloc(&s->d_colpart2, nstrips * (size_t)s->n * sizeof(double)));
    HIPCHK(hipMemsetAsync(s->d_rowpart2, 0, nsegs * (size_t)s->n * sizeof(double), s->stream));
    HIPCHK(hipMemsetAsync(s->d_colpart2, 0, nstrips * (size_t)s->n * sizeof(double), s->stream));
    return 0;
}

int queue_run_overlapped(ellhip_space* s, long long first, long long count) {
    int rc = overlap_setup(s);
    if (rc) return rc;
    SideGuard guard{s};
    for (long long i = first; i < first + count; ++i) {
        rc = queue_prime_impl(s, i);  // (only the first cut of a run pays its GEMV on the main stream)
        if (rc) return rc;
        const long long next = (i + 1 < s->qk) ? i + 1 : -1;
        // this cut becomes recorded update number npend; when that fills the slots an apply pass follows it, and the
        // next GEMV has to read what that pass writes
        const bool ahead = next >= 0 && symv_ok(s) && s->npend + 1 < s->defer;
        const int set = s->part_set ^ 1;
        if (ahead) {
            rc = side_fork(s);
            if (!rc) rc = launch_symv_tiles(s, qgrad(s, next), side_stream(s), set);
            if (!rc) rc = side_mark(s);
            if (rc) return rc;
        }
        rc = queue_cut_impl(s, i);
        if (rc) return rc;
        if (!ahead) {
            rc = queue_commit_impl(s, i, next);  // apply pass if due, then GEMV + reduction on the main stream
            if (rc) return rc;
            continue;
        }
        // what queue_commit_impl does on this schedule when no apply pass is due, with the GEMV already in flight
        s->shrink_pending = false;
        s->dots_np = 0;
        s->dots_need_gy = false;
        rc = side_join(s);
        if (rc) return rc;
        s->part_set = set;
        rc = launch_symv_reduce(s, qgrad(s, next), s->d_gt[s->cur ^ 1]);
        if (rc) return rc;
        s->cur ^= 1;
        s->primed = true;
        s->g_cur = qgrad(s, next);
        s->primed_qindex = next;
    }
    return 0;
}

// ---- pipelined queue runs with several cuts' GEMVs in one pass over Q_base (ELLHIP_OPT_LOOKAHEAD) ---------------------
// The same observation taken further: between two apply passes EVERY queued cut's y = Q_base g refers to the same
// matrix, so the products of a group of L consecutive queued cuts are formed in ONE pass over the lower triangle
// (k_symv_multi: each element loaded once, used for L gradients) -- (4 / L) n^2 bytes per update instead of 4 n^2.
// The reductions and scalar stages then run cut by cut as before (cut l's correction needs the vector cut l - 1
// recorded).  A group never reaches across an apply pass or the end of the run; a cut that arrives primed (by an
// earlier call) is taken by itself first.  Two kernels:
//   lookahead <= 3   k_symv_multi on the vector ALU: per vector k_symv's arithmetic, bit-identical results;
//   lookahead >  3   k_symm_mfma on the FP64 matrix cores (n a multiple of 64): up to 16 gradients in the time of ONE
//                    pass; y differs from k_symv's by a few ulp (own association, fused multiply-add) -- inside the
//                    contract's 1e-10, not bit-identical to the other schedules.
bool multi_ok(const ellhip_space* s) {
    return s->lookahead > 1 && s->variant == ELLHIP_SPACE_ELL && !s->sharded && symv_ok(s);
}
bool multi_mfma(const ellhip_space* s) { return s->lookahead > MULTI_VALU_MAX && (s->n % 64) == 0; }
// a symmetric row shard: the matrix-core groups only, and every cut through them (a single cut's GEMV would need the
// owner's per-cut collective in between); its owner calls queue_run_multi with grp_exchange set
bool multi_shard_ok(const ellhip_space* s) {
    return s->variant == ELLHIP_SPACE_ELL && s->sharded && s->shard_symmetric && symv_ok(s) && multi_mfma(s) &&
           (s->row0 % 64) == 0 && ((s->row0 + s->nrows) % 64 == 0 || s->row0 + s->nrows == s->n);
}

constexpr int MULTI_NO_MEMORY = 1;  // multi_setup: the buffers do not fit; the handle has been switched to lookahead 1

void multi_free(ellhip_space* s) {
    double** bufs[] = {&s->d_rowpart_m, &s->d_colpart_m, &s->d_gT, &s->d_grpY, &s->d_gpart, &s->d_cpart, &s->d_gsums};
    for (double** b : bufs) {
        if (*b) (void)hipFree(*b);
        *b = nullptr;
    }
    if (s->d_gout) (void)hipFree(s->d_gout);
    s->d_gout = nullptr;
    if (s->d_symm_tiles) (void)hipFree(s->d_symm_tiles);
    s->d_symm_tiles = nullptr;
    if (s->d_symm_queue) (void)hipFree(s->d_symm_queue);
    s->d_symm_queue = nullptr;
    s->symm_ntiles = 0;
}

// The group runs' buffers: 2 x 32 sets of partial sums (n^2 / 64 + n^2 / 2048 doubles each) are about n^2 doubles beside
// the matrix' n^2 (4.4 GB at n = 32768).  When they do not fit, the handle continues with one product per pass
// (returns MULTI_NO_MEMORY once; ELLHIP_OPT_LOOKAHEAD reads 1 afterwards): slower, not an error.
int multi_setup(ellhip_space* s) {
    if (s->d_rowpart_m) return 0;
    const size_t nb = (size_t)((s->n + 127) / 128);
    // two halves of MULTI_MAX sets each (and of the packed gradients): the next group's products are formed on the second
    // stream while this group's stage reads the other half
    const size_t rbytes = (size_t)2 * MULTI_MAX * rowpart_elems(s) * sizeof(double);
    const size_t cbytes = (size_t)2 * MULTI_MAX * colpart_elems(s) * sizeof(double);
    hipError_t e = hipMalloc(&s->d_rowpart_m, rbytes);
    if (e == hipSuccess) e = hipMalloc(&s->d_colpart_m, cbytes);
    if (e == hipSuccess) e = hipMalloc(&s->d_gT, (size_t)2 * s->n * MULTI_MAX * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&s->d_grpY, (size_t)GRP_MAX * (size_t)s->n * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&s->d_gpart, (size_t)GRP_MAX * nb * (MAXPEND + 1) * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&s->d_cpart, nb * GRP_MAX * GRP_MAX * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&s->d_gout, sizeof(GroupOut));
    if (e == hipSuccess) e = hipMalloc(&s->d_gsums, (size_t)(GRP_MAX * (MAXPEND + 1) + GRP_MAX * GRP_MAX) * sizeof(double));
    // the matrix-core pass draws its tiles from a queue, largest first (k_symm_mfma_q)
    std::vector<SymmTile> tiles;
    if ((s->n % 64) == 0 && (s->row0 % 64) == 0) {  // (whatever the lookahead is right now: the option may change)
        const long long seg = s->symv_seg;
        const long long nstrips = (s->nrows + SYMV_H - 1) / SYMV_H, nsegs = (s->n + seg - 1) / seg;
        auto blocks_of = [&](const SymmTile& t) {
            const long long r0 = s->row0 + (long long)t.I * SYMV_H, c0 = (long long)t.J * seg;
            return (std::min(c0 + seg, r0 + SYMV_H) - c0) / 16;
        };
        for (long long I = nstrips - 1; I >= 0; --I)
            for (long long J = 0; J < nsegs; ++J)
                if (J * seg <= s->row0 + I * SYMV_H + SYMV_H - 1) tiles.push_back({(int)I, (int)J});
        std::stable_sort(tiles.begin(), tiles.end(), [&](const SymmTile& a, const SymmTile& b) { return blocks_of(a) > blocks_of(b); });
        if (e == hipSuccess) e = hipMalloc(&s->d_symm_tiles, tiles.size() * sizeof(SymmTile));
        if (e == hipSuccess) e = hipMalloc(&s->d_symm_queue, 256);
    }
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        multi_free(s);
        s->lookahead = 1;
        return MULTI_NO_MEMORY;
    }
    if (e != hipSuccess) {
        multi_free(s);
        return fail(ELLHIP_E_HIP, "hipMalloc (group-run buffers)", e);
    }
    HIPCHK(hipMemsetAsync(s->d_rowpart_m, 0, rbytes, s->stream));
    HIPCHK(hipMemsetAsync(s->d_colpart_m, 0, cbytes, s->stream));
    HIPCHK(hipMemsetAsync(s->d_gout, 0, sizeof(GroupOut), s->stream));
    if (!tiles.empty()) {
        HIPCHK(hipMemcpyAsync(s->d_symm_tiles, tiles.data(), tiles.size() * sizeof(SymmTile), hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipMemsetAsync(s->d_symm_queue, 0, 256, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));  // (`tiles` is pageable host memory going out of scope)
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, s->device));
        s->symm_ntiles = (int)tiles.size();
        // two workgroups per CU is what the registers allow; more would only wait for the first ones to drain the queue
        s->symm_wgs = (int)std::min<size_t>(tiles.size(), (size_t)2 * (size_t)prop.multiProcessorCount);
        // The runtime loads a kernel at its first launch (~250 us for these): a short queue run must not meet that inside its
        // own 20 cuts because the runs before it happened to stay below 17 cuts per group.  One empty launch each (no tiles).
        const bool nt = s->sh_gemv.nt != 0, wide = s->symv_seg == SYMV_SEG;
#define ELLHIP_SYMM_WARM(...)                                                                                                        \
    hipLaunchKernelGGL((__VA_ARGS__), dim3(1), dim3(256), 0, s->stream, (const double*)s->d_Q, s->ld, s->n, s->row0,                 \
                       (const double*)s->d_gT, 0, s->d_rowpart_m, s->d_colpart_m, (long long)rowpart_elems(s),                      \
                       (long long)colpart_elems(s), (const DevState*)s->d_st, (const SymmTile*)s->d_symm_tiles, 0, s->d_symm_queue)
        if (wide && nt) {
            ELLHIP_SYMM_WARM(k_symm_mfma_q<true, SYMV_SEG>);
            ELLHIP_SYMM_WARM(k_symm_mfma_q2<true, SYMV_SEG>);
        } else if (wide) {
            ELLHIP_SYMM_WARM(k_symm_mfma_q<false, SYMV_SEG>);
            ELLHIP_SYMM_WARM(k_symm_mfma_q2<false, SYMV_SEG>);
        } else if (nt) {
            ELLHIP_SYMM_WARM(k_symm_mfma_q<true, SYMV_SEG_SMALL>);
            ELLHIP_SYMM_WARM(k_symm_mfma_q2<true, SYMV_SEG_SMALL>);
        } else {
            ELLHIP_SYMM_WARM(k_symm_mfma_q<false, SYMV_SEG_SMALL>);
            ELLHIP_SYMM_WARM(k_symm_mfma_q2<false, SYMV_SEG_SMALL>);
        }
#undef ELLHIP_SYMM_WARM
        HIPCHK(hipGetLastError());
    }
    return 0;
}

// the scalar stage of a group whose products sit in the partial-sum sets 2 .. 2 + g - 1 (group_kernels.hpp)
// phase 0: the reductions (they read the pass' partial sums at HBM rate and want the whole card); phase 1: the rest (Gram slices,
// sums, the one-wave recurrence, the vectors: short kernels that run beside the NEXT group's pass on the CU slots it leaves free)
template <int NP>
int group_stage_go(ellhip_space* s, long long i, int g, int half, int phase) {
    const unsigned nb = (unsigned)((s->n + 127) / 128);
    const double* grads = qgrad(s, i);
    if (phase == 0) {
        ProfScope ps(s, CLS_SYMV_REDUCE);
        const double* rowp = s->d_rowpart_m + (size_t)half * MULTI_MAX * rowpart_elems(s);
        const double* colp = s->d_colpart_m + (size_t)half * MULTI_MAX * colpart_elems(s);
        if (s->sharded)
            hipLaunchKernelGGL(k_group_reduce<0>, dim3(nb, (unsigned)g), dim3(256), 0, s->stream, s->n, s->row0, s->nrows,
                               (long long)s->symv_seg, rowp, colp, (long long)rowpart_elems(s), (long long)colpart_elems(s),
                               s->d_grpY, grads, s->n, (const double*)s->d_pend, s->d_gpart, (const DevState*)s->d_st);
        else
            hipLaunchKernelGGL(k_group_reduce<NP>, dim3(nb, (unsigned)g), dim3(256), 0, s->stream, s->n, s->row0, s->nrows,
                               (long long)s->symv_seg, rowp, colp, (long long)rowpart_elems(s), (long long)colpart_elems(s),
                               s->d_grpY, grads, s->n, (const double*)s->d_pend, s->d_gpart, (const DevState*)s->d_st, s->npend);
        HIPCHK(hipGetLastError());
        if (!s->sharded) return 0;
    }
    if (phase == 0) {
        // the shards' partial products become the products: ONE collective for the whole group, then the dot products from
        // the complete vectors (every rank forms the same ones)
        if (!s->grp_exchange) return fail(ELLHIP_E_STATE, "group run of a row shard without the owner's collective");
        const int xrc = s->grp_exchange(s->grp_exchange_ctx, s->d_grpY, (long long)g * s->n, s->stream);
        if (xrc) return xrc;
        ProfScope ps(s, CLS_SYMV_REDUCE);
        hipLaunchKernelGGL(k_group_dots<NP>, dim3(nb, (unsigned)g), dim3(256), 0, s->stream, s->n, (const double*)s->d_grpY, grads,
                           s->n, (const double*)s->d_pend, s->d_gpart, (const DevState*)s->d_st, s->npend);
        HIPCHK(hipGetLastError());
        return 0;
    }
    ProfScope ps(s, CLS_SCALAR);
    hipLaunchKernelGGL(k_group_gram, dim3(nb), dim3(256), 0, s->stream, s->n, g, (const double*)s->d_grpY, grads, s->n,
                       s->d_cpart, (const DevState*)s->d_st);
    hipLaunchKernelGGL(k_group_sums<NP>, dim3((unsigned)g + GRP_MAX * GRP_MAX / 256), dim3(256), 0, s->stream, g, (int)nb, (const double*)s->d_gpart,
                       (const double*)s->d_cpart, s->d_gsums, (const DevState*)s->d_st);
    hipLaunchKernelGGL(k_group_scalar<NP>, dim3(1), dim3(64), 0, s->stream, g, (const double*)s->d_gsums, s->d_cpend, s->d_st,
                       EllCalcDev::make(s->n, s->use_parallel_cut), (const CutParams*)(s->d_qparams + i), s->npend,
                       s->d_qstatus + i, s->d_qtsq + i, s->d_gout);
    hipLaunchKernelGGL(k_group_apply<NP>, dim3((unsigned)((s->n + 255) / 256)), dim3(256), 0, s->stream, s->n, g,
                       (const double*)s->d_grpY, s->d_pend, s->d_xc, s->npend, (const GroupOut*)s->d_gout);
    HIPCHK(hipGetLastError());
    return 0;
}

template <int SEG>
void symm_mfma_go(ellhip_space* s, const double* g_dev, int lv, hipStream_t st, int half, int wgs) {
    double* gT = s->d_gT + (size_t)half * (size_t)s->n * MULTI_MAX;
    double* rowp = s->d_rowpart_m + (size_t)half * MULTI_MAX * rowpart_elems(s);
    double* colp = s->d_colpart_m + (size_t)half * MULTI_MAX * colpart_elems(s);
    unsigned* queue = s->d_symm_queue + 32 * half;
    const int nvw = lv > SMM_NV ? SMM_NV2 : SMM_NV;  // one or two 16-wide column tiles
    hipLaunchKernelGGL(k_pack_grads, dim3((unsigned)((s->n * nvw + 255) / 256)), dim3(256), 0, st, g_dev, s->n, lv, s->n, gT, queue, nvw);
    const bool nt = s->sh_gemv.nt != 0;
#define ELLHIP_SYMM_Q(...)                                                                                                          \
    hipLaunchKernelGGL((__VA_ARGS__), dim3((unsigned)wgs), dim3(256), 0, st, (const double*)s->d_Q, s->ld, s->n, s->row0,                \
                       (const double*)gT, lv, rowp, colp, (long long)rowpart_elems(s), (long long)colpart_elems(s),                 \
                       (const DevState*)s->d_st, (const SymmTile*)s->d_symm_tiles, s->symm_ntiles, queue)
    if (nvw == SMM_NV) {
        if (nt) ELLHIP_SYMM_Q(k_symm_mfma_q<true, SEG>);
        else ELLHIP_SYMM_Q(k_symm_mfma_q<false, SEG>);
    } else {
        if (nt) ELLHIP_SYMM_Q(k_symm_mfma_q2<true, SEG>);
        else ELLHIP_SYMM_Q(k_symm_mfma_q2<false, SEG>);
    }
#undef ELLHIP_SYMM_Q
}
// How many of the `rem` cuts up to the next apply pass (or the end of the run) go into the next group, `cap` = what the handle's
// lookahead and the kernels allow.  A matrix-core pass costs about the same for 4 gradients as for 16 (0.27 ms at n = 16384) and
// 0.40-0.45 for 17-32 (two column tiles over the same block of Q), so: as many as fit; and where only the 16-wide pass is allowed
// and more than one but less than two full groups are left, two even groups rather than a full one and a small one (the first
// group's stage then overlaps a pass that carries its share).
long long group_size(const ellhip_space* s, long long cap, long long rem) {
    long long g = std::min(cap, rem);
    if (multi_mfma(s) && cap <= SMM_NV && rem > cap && rem < 2 * cap) g = (rem + 1) / 2;
    return g;
}

// beside: the pass is issued next to the previous group's stage -- it leaves a sixteenth of its workgroup slots free, where that
// stage's short kernels run (a pass that draws its tiles from a queue keeps every slot it gets until it ends; with a 32nd free
// k_group_sums did not get on the card before the pass ended, with an eighth the pass itself lost more than the stage gained)
int symm_go(ellhip_space* s, const double* g_dev, int lv, hipStream_t st, int half, bool beside = false) {
    ProfScope ps(s, CLS_SYMV, st);
    const int wgs = beside ? std::max(1, s->symm_wgs - std::max(1, s->symm_wgs / 16)) : s->symm_wgs;
    if (s->symv_seg == SYMV_SEG) symm_mfma_go<SYMV_SEG>(s, g_dev, lv, st, half, wgs);
    else symm_mfma_go<SYMV_SEG_SMALL>(s, g_dev, lv, st, half, wgs);
    HIPCHK(hipGetLastError());
    return 0;
}

// (the thread-to-element map, and with it the bits, follow the handle's segment width and are independent of the rows
// in flight: k_symv<8, .., 512> of the narrow segments and k_symv_multi<2, .., 512, LV> give the same sums)
template <int SEG, int LV>
void symv_multi_go(ellhip_space* s, const double* g_dev) {
    const unsigned nstrips = (unsigned)((s->nrows + SYMV_H - 1) / SYMV_H);
    const unsigned nsegs = (unsigned)((s->n + SEG - 1) / SEG);
    if (s->sh_gemv.nt != 0)
        hipLaunchKernelGGL((k_symv_multi<2, true, SEG, LV>), dim3(nstrips, nsegs), dim3(256), 0, s->stream,
                           (const double*)s->d_Q, s->ld, s->n, s->row0, s->nrows, g_dev, s->n, s->d_rowpart_m, s->d_colpart_m,
                           (long long)rowpart_elems(s), (long long)colpart_elems(s), (const DevState*)s->d_st);
    else
        hipLaunchKernelGGL((k_symv_multi<2, false, SEG, LV>), dim3(nstrips, nsegs), dim3(256), 0, s->stream,
                           (const double*)s->d_Q, s->ld, s->n, s->row0, s->nrows, g_dev, s->n, s->d_rowpart_m, s->d_colpart_m,
                           (long long)rowpart_elems(s), (long long)colpart_elems(s), (const DevState*)s->d_st);
}

int queue_run_multi(ellhip_space* s, long long first, long long count) {
    int rc = multi_setup(s);
    if (rc) return rc;  // (MULTI_NO_MEMORY: the caller goes on with the schedules that need no extra buffers)
    const long long end = first + count;
    if (s->sharded && (s->primed || !s->grp_exchange))
        return fail(ELLHIP_E_STATE, "group run of a row shard: the owner takes a primed cut by itself and supplies the collective");
    // Inside this run the group stage may let more updates pile up than the handle's depth (its kernels are sized for
    // MAXPEND = 48; the per-cut kernels of every other path for the depth): half as many apply passes.  Whatever leaves
    // this function has fewer recorded than the depth again.
    const bool deep_ok = multi_mfma(s) && s->defer == 24 && s->apply_lower && s->queue_depth > s->defer;
    const int qdepth = deep_ok ? s->queue_depth : s->defer;
    if (qdepth != 8 && qdepth != 16 && qdepth != 24 && qdepth != MAXPEND)  // (the group kernels' slot counts; pend / cpend hold MAXPEND)
        return fail(ELLHIP_E_STATE, "queue run: recorded-update depth must be 8, 16, 24 or 48");
    // (round 3, the pass beside the whole stage: +1.5 % / +8 % at n = 16384 for 200 / 20 cuts per run, -3 % at n = 32768, so the
    // second stream stopped at n = 24576.  Round 4, the pass behind the stage's reductions and beside the rest of it on the slots
    // it leaves free: +6 % at n = 16384 (200 cuts), +1.5 % at n = 32768 -- used at every size)
    const bool use_side = multi_mfma(s) && s->overlap != 0 && !s->sharded;
    if (use_side) {
        rc = side_setup(s);
        if (rc) return rc;
    }
    SideGuard guard{s};
    int half = 0;                       // the half of the partial-sum sets the current group's products are in
    long long ahead_i = -1, ahead_g = 0;  // the group whose products are in flight on the second stream
    long long i = first;
    while (i < end) {
        long long g = 1;
        if (!(s->primed && s->primed_qindex == i)) {
            rc = ensure_committed(s);
            if (rc) return rc;
            const long long room = (long long)qdepth - s->npend;  // cuts that can still be recorded before the apply pass
            const long long cap = std::min<long long>(s->lookahead, multi_mfma(s) ? MULTI_MAX : MULTI_VALU_MAX);
            const long long rem = std::min(end - i, room);  // cuts up to the next apply pass or the end of the run
            g = group_size(s, cap, rem);
        }
        if (g <= 1 && !s->sharded) {  // a cut primed earlier, the last one before an apply pass, the last one of the run
            if (s->npend >= s->defer) {  // (the per-cut kernels hold `depth` recorded updates)
                rc = flush_pending(s, nullptr, nullptr);
                if (rc) return rc;
            }
            rc = queue_prime_impl(s, i);
            if (!rc) rc = queue_cut_impl(s, i);
            if (!rc) rc = queue_commit_impl(s, i, -1);
            if (rc) return rc;
            i += 1;
            continue;
        }
        if (multi_mfma(s)) {
            const long long cap = MULTI_MAX;
            rc = side_join(s);  // (this group's products, when they were issued ahead)
            if (rc) return rc;
            if (!(ahead_i == i && ahead_g == g)) {  // this group's products are not in the sets yet
                rc = symm_go(s, qgrad(s, i), (int)g, s->stream, half);
                if (rc) return rc;
            }
            ahead_i = -1;
            // the NEXT group's products on the second stream beside this group's stage -- when there is one inside this run
            // and no apply pass comes first (it would read the matrix that pass writes)
            const long long i2 = i + g, room2 = (long long)qdepth - (s->npend + g);
            long long g2 = 0;
            if (i2 < end && room2 > 0) {
                const long long cap2 = std::min<long long>(s->lookahead, cap), rem2 = std::min(end - i2, room2);
                g2 = group_size(s, cap2, rem2);  // (the rule the next trip of the loop applies)
            }
            // the whole group's scalar stage: the cuts' omegas and coefficients from dot products that exist when the group
            // starts, then the vectors in one elementwise pass.  First its reductions (they stream the pass' partial sums and want
            // the whole card) ...
            drop_prime(s);
            s->dots_np = 0;
            s->xc_host_valid = false;
#define ELLHIP_STAGE(PHASE)                                                                                          \
    (qdepth == MAXPEND ? group_stage_go<MAXPEND>(s, i, (int)g, half, PHASE)                                          \
     : qdepth == 24    ? group_stage_go<24>(s, i, (int)g, half, PHASE)                                               \
     : qdepth == 16    ? group_stage_go<16>(s, i, (int)g, half, PHASE) : group_stage_go<8>(s, i, (int)g, half, PHASE))
            rc = ELLHIP_STAGE(0);
            if (rc) return rc;
            // ... then the NEXT group's products on the second stream, behind those reductions and beside the rest of this
            // stage (short kernels, one of them a single wave: they run on the slots the pass leaves free)
            if (use_side && g2 >= 2) {
                rc = side_fork(s);  // after this group's reductions: the last reader of the other half, every writer of Q so far
                if (!rc) rc = symm_go(s, qgrad(s, i2), (int)g2, side_stream(s), half ^ 1, true);
                if (!rc) rc = side_mark(s);
                if (rc) return rc;
                ahead_i = i2;
                ahead_g = g2;
            }
            rc = ELLHIP_STAGE(1);
#undef ELLHIP_STAGE
            if (rc) return rc;
            s->npend += (int)g;       // (optimistic, as after every queue cut: ellhip_queue_results settles it after a halt)
            s->scalars_stale = true;
            s->shrink_pending = false;
            if (s->npend >= qdepth) {
                rc = flush_pending(s, nullptr, nullptr);
                if (rc) return rc;
            }
            if (ahead_i >= 0) half ^= 1;
            i += g;
            continue;
        }
        {
            ProfScope ps(s, CLS_SYMV);
            const bool wide = s->symv_seg == SYMV_SEG;
            if (g == 2) {
                if (wide) symv_multi_go<SYMV_SEG, 2>(s, qgrad(s, i));
                else symv_multi_go<SYMV_SEG_SMALL, 2>(s, qgrad(s, i));
            } else {
                if (wide) symv_multi_go<SYMV_SEG, 3>(s, qgrad(s, i));
                else symv_multi_go<SYMV_SEG_SMALL, 3>(s, qgrad(s, i));
            }
            HIPCHK(hipGetLastError());
        }
        for (long long l = 0; l < g; ++l) {
            s->part_set = 2 + (int)l;
            s->dots_np = 0;
            s->dots_need_gy = false;
            rc = launch_symv_reduce(s, qgrad(s, i + l), s->d_gt[s->cur]);
            s->part_set = 0;
            if (rc) return rc;
            s->primed = true;
            s->g_cur = qgrad(s, i + l);
            s->primed_qindex = i + l;
            rc = queue_cut_impl(s, i + l);
            if (!rc) rc = queue_commit_impl(s, i + l, -1);  // (the apply pass when this cut fills the slots)
            if (rc) return rc;
        }
        i += g;
    }
    if (s->npend >= s->defer) return flush_pending(s, nullptr, nullptr);
    return 0;
}

// ---- queue runs with the matrix parked on-chip (resident_kernels.hpp) ------------------------------------------------
// Chosen for a run of at least RS_MIN_COUNT queued cuts of an unsharded Ell handle whose lower triangle fits the
// register files: R = the smallest super-tile edge with no more super-tiles than the device has CUs, and one workgroup
// of k_ell_resident<R> must fit a CU.  Not while no_defer_trick is set (the kernel has no scale pass) or before a
// non-symmetric input has been mirrored, and not for the device-resident cutting-plane loops (tolerance stop).
constexpr long long RS_MIN_COUNT = 4;

int resident_setup(ellhip_space* s) {  // once per handle: can this n run resident on this device, and with which R?
    if (s->rs_R != 0) return 0;
    s->rs_R = -1;
    if (s->variant != ELLHIP_SPACE_ELL || s->sharded) return 0;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, s->device));
    const int T = (int)((s->n + RS_TS - 1) / RS_TS);
    for (int r = 1; r <= RS_RMAX; ++r) {
        const int S = (T + r - 1) / r;
        if (S > RS_SMAX || S * (S + 1) / 2 > prop.multiProcessorCount) continue;
        int occ = 0;
        const void* k = r == 1 ? (const void*)k_ell_resident<1> : (r == 2 ? (const void*)k_ell_resident<2> : (const void*)k_ell_resident<3>);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, RS_THREADS, 0) != hipSuccess || occ < 1) continue;
        const size_t G = (size_t)S * (S + 1) / 2, NV = 2 * (size_t)r * RS_TS;
        HIPCHK(hipMalloc(&s->d_rs_part, 2 * G * NV * sizeof(double)));
        HIPCHK(hipMalloc(&s->d_rs_omega, 2 * G * sizeof(double)));
        HIPCHK(hipMalloc(&s->d_rs_bar, RS_BAR_WORDS * sizeof(unsigned)));
        HIPCHK(hipMalloc(&s->d_rs_xc0, (size_t)s->n * sizeof(double)));
        HIPCHK(hipMalloc(&s->d_rs_st0, sizeof(DevState)));
        s->rs_R = r;
        s->rs_S = S;
        break;
    }
    return 0;
}

bool resident_ok(ellhip_space* s, long long count) {
    if (!s->resident || count < RS_MIN_COUNT || s->variant != ELLHIP_SPACE_ELL || s->sharded || s->no_defer_trick || s->needs_mirror)
        return false;
    if (s->h_result && s->h_result->tol >= 0.0) return false;
    if (resident_setup(s) != 0) return false;
    return s->rs_R > 0;
}

// (one resident grid per device at a time within this process, and none beside an EllStable persistent solve: CoresScope)
constexpr int RS_FALLBACK = 1;  // resident_run: the batch did not run (or was abandoned and undone): take the streamed schedule

// One batch = one cooperative launch.  The call returns when the batch has finished (one stream synchronisation per
// batch: the verdict "committed or abandoned" is needed before anything else may be enqueued behind it).
//   0            the batch ran: Q (lower triangle), xc, DevState and the queue results are those of `count` more cuts;
//   RS_FALLBACK  nothing happened as far as the caller can tell -- the launch was refused (the grid cannot be co-resident
//                on this device as it is partitioned now) or a bounded wait inside the kernel gave up, in which case no
//                tile was written back (rs_commit) and xc / DevState / the batch's queue results have been restored from
//                the snapshots taken below.  The caller continues with the streamed schedule; so does the handle from
//                now on (s->resident = 0).
int resident_run(ellhip_space* s, long long first, long long count) {
    // the lower triangle must be current: commit a pending shrink, apply what the recorded schedule holds; a primed
    // gradient is simply dropped (the kernel forms every Q g itself)
    int rc = ensure_committed(s);
    if (rc) return rc;
    if (s->npend > 0) {
        rc = flush_pending(s, nullptr, nullptr);
        if (rc) return rc;
    }
    drop_prime(s);
    s->xc_host_valid = false;
    ResidentArgs A{};
    A.Q = s->d_Q;
    A.ld = s->ld;
    A.n = s->n;
    A.T = (int)((s->n + RS_TS - 1) / RS_TS);
    A.R = s->rs_R;
    A.S = s->rs_S;
    A.qgrads = s->d_qgrads;
    A.qparams = s->d_qparams;
    A.qstatus = s->d_qstatus;
    A.qtsq = s->d_qtsq;
    A.first = first;
    A.count = count;
    A.xc = s->d_xc;
    A.st = s->d_st;
    A.part = s->d_rs_part;
    A.omega_part = s->d_rs_omega;
    A.ctr = s->d_rs_bar;
    A.stamps = nullptr;
    A.calc = EllCalcDev::make(s->n, s->use_parallel_cut);
    A.fault_at = s->rs_fault_at;
    const unsigned G = (unsigned)(s->rs_S * (s->rs_S + 1) / 2);
    CoresScope alone(s);  // (held until the batch has finished: the function synchronises the stream before it returns)
    hipLaunchKernelGGL(k_rs_prepare, dim3((unsigned)std::max<long long>(1, std::min<long long>(32, s->n / 256))), dim3(256), 0, s->stream,
                       (const double*)s->d_xc, s->d_rs_xc0, s->n, (const DevState*)s->d_st, s->d_rs_st0, s->d_rs_bar);
    HIPCHK(hipGetLastError());
    {
        ProfScope ps(s, CLS_RESIDENT);
        const void* k = s->rs_R == 1 ? (const void*)k_ell_resident<1> : (s->rs_R == 2 ? (const void*)k_ell_resident<2> : (const void*)k_ell_resident<3>);
        void* args[] = {&A};
        // ELLHIP_OPT_RESIDENT = 1 (default): a cooperative launch -- the runtime refuses a grid that cannot be co-resident and
        // runs one cooperative grid at a time, across processes too; measured at the plain launch's speed (n = 4096, 200 cuts
        // per batch: 1.84 vs 1.87 ms).  2: a plain launch, for a stack without cooperative launches.  Either way the batches
        // of this process are serialised per device by CoresScope (held here until the batch has finished),
        // and whatever still keeps workgroups out ends in the bounded waits giving up: the batch is abandoned as a whole
        // and rerun on the streamed schedule (below).
        hipError_t e;
        if (s->resident == 1) {
            e = hipLaunchCooperativeKernel(k, dim3(G), dim3(RS_THREADS), args, 0, s->stream);
        } else {
            e = hipLaunchKernel(k, dim3(G), dim3(RS_THREADS), args, 0, s->stream);
        }
        if (e != hipSuccess) {
            // not co-resident on this device (partitioned, CU mask, ...): no resident batches on this handle any more
            (void)hipGetLastError();
            s->resident = 0;
            return RS_FALLBACK;
        }
    }
    HIPCHK(hipMemcpyAsync(s->h_result, s->d_st, sizeof(DevState), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (s->h_result->solve_err) {
        // abandoned: Q was not written.  Put back what the kernel's cuts wrote on the way -- xc (the diagonal workgroups),
        // the scalar state, the batch's queue results ("not run yet") -- and let the streamed schedule take the batch.
        HIPCHK(hipMemcpyAsync(s->d_xc, s->d_rs_xc0, (size_t)s->n * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->d_st, s->d_rs_st0, sizeof(DevState), hipMemcpyDeviceToDevice, s->stream));
        HIPCHK(hipMemsetAsync(s->d_qstatus + first, 0xff, (size_t)count * sizeof(int), s->stream));
        HIPCHK(hipMemsetAsync(s->d_qtsq + first, 0, (size_t)count * sizeof(double), s->stream));
        HIPCHK(hipMemcpyAsync(s->h_result, s->d_st, sizeof(DevState), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        s->resident = 0;
        s->rs_abandoned += 1;
        (void)fail(0, "a resident batch was abandoned (bounded in-launch wait) and rerun on the streamed schedule; "
                      "ELLHIP_OPT_RESIDENT is now 0 on this handle");
        return RS_FALLBACK;
    }
    // The kernel wrote the lower triangle (diagonal tiles whole).  Where the streamed schedule of this handle reads
    // full rows, the mirrored half is rebuilt at once; a handle on the lower-triangle schedule leaves it stale as its
    // own apply passes do (make_q_current mirrors before anything observes Q).
    if (symv_ok(s) && s->apply_lower) {
        s->upper_stale = true;
    } else {
        const unsigned t = (unsigned)((s->n + 31) / 32);
        hipLaunchKernelGGL(k_mirror_lower_now, dim3(t, t), dim3(256), 0, s->stream, s->d_Q, s->ld, s->n);
        HIPCHK(hipGetLastError());
        s->upper_stale = false;
    }
    s->shrink_pending = false;
    s->kappa = s->h_result->kappa;
    s->tsq = s->h_result->tsq;
    s->scalars_stale = false;
    return 0;
}

}  // namespace

// ================================================================================= C ABI ======

extern "C" {

int ellhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* ellhip_last_error(void) { return g_last_error.c_str(); }

const char* ellhip_version(void) { return "ellhip 0.4.0 gfx950"; }

int ellhip_create(ellhip_space** out, int variant, int64_t n, double kappa, const double* mq, const double* diag,
                  const double* xc, int device) {
    return create_impl(out, variant, n, 0, n, false, kappa, mq, diag, xc, device);
}

int ellhip_create_shard(ellhip_space** out, int64_t n, int64_t row0, int64_t nrows, double kappa,
                        const double* mq_rows, const double* diag, const double* xc, int device) {
    return create_impl(out, ELLHIP_SPACE_ELL, n, row0, nrows, true, kappa, mq_rows, diag, xc, device);
}

void ellhip_destroy(ellhip_space* s) {
    if (!s) return;
    DeviceGuard guard(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (s->aux_stream) (void)hipStreamSynchronize(s->aux_stream);  // nothing may be in flight when the buffers go
    if (s->symv_stream) (void)hipStreamSynchronize(s->symv_stream);
    for (auto& pe : s->prof_events) {
        (void)hipEventDestroy(pe.a);
        (void)hipEventDestroy(pe.b);
    }
    queue_free(s);
    if (s->d_Q) (void)hipFree(s->d_Q);
    if (s->d_xc) (void)hipFree(s->d_xc);
    for (int k = 0; k < 2; ++k) {
        if (s->d_stage[k]) (void)hipFree(s->d_stage[k]);
        if (s->d_gt_own[k]) (void)hipFree(s->d_gt_own[k]);
        if (s->h_stage[k]) (void)hipHostFree(s->h_stage[k]);
    }
    if (s->d_work) (void)hipFree(s->d_work);
    if (s->d_hpart) (void)hipFree(s->d_hpart);
    if (s->d_stpend) (void)hipFree(s->d_stpend);
    if (s->d_rbuf) (void)hipFree(s->d_rbuf);
    if (s->d_wkeep) (void)hipFree(s->d_wkeep);
    if (s->d_fnext) (void)hipFree(s->d_fnext);
    if (s->d_ftiles16) (void)hipFree(s->d_ftiles16);
    if (s->d_qhpart) (void)hipFree(s->d_qhpart);
    if (s->d_partial) (void)hipFree(s->d_partial);
    if (s->d_pend) (void)hipFree(s->d_pend);
    if (s->d_cpend) (void)hipFree(s->d_cpend);
    if (s->d_rs_part) (void)hipFree(s->d_rs_part);
    if (s->d_rs_omega) (void)hipFree(s->d_rs_omega);
    if (s->d_rs_bar) (void)hipFree(s->d_rs_bar);
    if (s->d_rs_xc0) (void)hipFree(s->d_rs_xc0);
    if (s->d_rs_st0) (void)hipFree(s->d_rs_st0);
    if (s->d_rowpart) (void)hipFree(s->d_rowpart);
    if (s->d_colpart) (void)hipFree(s->d_colpart);
    if (s->d_rowpart2) (void)hipFree(s->d_rowpart2);
    if (s->d_colpart2) (void)hipFree(s->d_colpart2);
    if (s->d_rowpart_m) (void)hipFree(s->d_rowpart_m);
    if (s->d_colpart_m) (void)hipFree(s->d_colpart_m);
    if (s->d_gT) (void)hipFree(s->d_gT);
    if (s->d_grpY) (void)hipFree(s->d_grpY);
    if (s->d_gpart) (void)hipFree(s->d_gpart);
    if (s->d_cpart) (void)hipFree(s->d_cpart);
    if (s->d_gout) (void)hipFree(s->d_gout);
    if (s->d_gsums) (void)hipFree(s->d_gsums);
    if (s->d_symm_tiles) (void)hipFree(s->d_symm_tiles);
    if (s->d_symm_queue) (void)hipFree(s->d_symm_queue);
    if (s->ev_symv) (void)hipEventDestroy(s->ev_symv);
    if (s->ev_side_go) (void)hipEventDestroy(s->ev_side_go);
    if (s->symv_stream) (void)hipStreamDestroy(s->symv_stream);
    if (s->d_flags) (void)hipFree(s->d_flags);
    if (s->d_st) (void)hipFree(s->d_st);
    if (s->h_result) (void)hipHostFree(s->h_result);
    if (s->h_live) (void)hipHostFree(s->h_live);
    if (s->h_xc) (void)hipHostFree(s->h_xc);
    if (s->d_pub_arrived) (void)hipFree(s->d_pub_arrived);
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    if (s->ev_join) (void)hipEventDestroy(s->ev_join);
    if (s->aux_stream) (void)hipStreamDestroy(s->aux_stream);
    if (s->own_stream) (void)hipStreamDestroy(s->own_stream);
    delete s;
}

int ellhip_clone(const ellhip_space* src_c, ellhip_space** out) {
    if (!src_c || !out) return fail(ELLHIP_E_INVALID, "NULL argument");
    *out = nullptr;
    ellhip_space* src = const_cast<ellhip_space*>(src_c);  // committing a pending shrink is not observable
    DeviceGuard guard(src->device);
    int rc = make_q_current(src);
    if (rc) return rc;
    ellhip_space* s = new (std::nothrow) ellhip_space();
    if (!s) return fail(ELLHIP_E_NOMEM, "host allocation failed");
    s->variant = src->variant;
    s->n = src->n;
    s->ld = src->ld;
    s->row0 = src->row0;
    s->nrows = src->nrows;
    s->sharded = src->sharded;
    s->device = src->device;
    s->no_defer_trick = src->no_defer_trick;
    s->use_parallel_cut = src->use_parallel_cut;
    s->needs_mirror = src->needs_mirror;
    s->kappa = src->kappa;
    s->tsq = src->tsq;
    s->sh_gemv = src->sh_gemv;
    s->sh_rank1 = src->sh_rank1;
    s->sh_fused = src->sh_fused;
    s->sh_apply = src->sh_apply;
    s->symv = src->symv;
    s->symv_rw = src->symv_rw;
    s->symv_min_n = src->symv_min_n;
    s->apply_lower = src->apply_lower;
    s->apply_kernel = src->apply_kernel;
    s->fuse_dots = src->fuse_dots;
    s->overlap = src->overlap;
    s->lookahead = src->lookahead;
    s->queue_depth = src->queue_depth;
    s->resident = src->resident;
    s->shard_symmetric = src->shard_symmetric;
    s->upper_stale = src->upper_stale;
    s->symv_seg = src->symv_seg;
    s->stable_solve = src->stable_solve;
    s->stable_factor = src->stable_factor;
    rc = alloc_common(s);
    if (rc) {
        ellhip_destroy(s);
        return rc;
    }
    // order the copies after everything the source has in flight
    hipError_t e = hipStreamSynchronize(src->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(s->d_Q, src->d_Q, (size_t)s->nrows * (size_t)s->ld * sizeof(double),
                           hipMemcpyDeviceToDevice, s->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(s->d_xc, src->d_xc, (size_t)s->n * sizeof(double), hipMemcpyDeviceToDevice, s->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s->d_st, src->d_st, sizeof(DevState), hipMemcpyDeviceToDevice, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e != hipSuccess) {
        ellhip_destroy(s);
        return fail(ELLHIP_E_HIP, "clone copy", e);
    }
    // The device state is the truth (the source's host cache lags behind asynchronous queue cuts); the clone has no
    // queue, so it does not inherit a halted one.
    rc = read_back(s);
    if (!rc) {
        s->h_result->halted = 0;
        s->h_result->halted_in = 0;
        s->h_result->stop = STOP_NONE;
        s->h_result->tol = -1.0;
        s->h_result->niter = 0;
        e = hipMemcpyAsync(s->d_st, s->h_result, sizeof(DevState), hipMemcpyHostToDevice, s->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        if (e != hipSuccess) rc = fail(ELLHIP_E_HIP, "clone state", e);
    }
    if (rc) {
        ellhip_destroy(s);
        return rc;
    }
    if (src->defer > 1) {
        rc = ellhip_set_defer_depth(s, src->defer);
        if (rc) {
            ellhip_destroy(s);
            return rc;
        }
    }
    *out = s;
    return 0;
}

// ---- pipelined primitives ---------------------------------------------------------------------

int ellhip_prime(ellhip_space* s, const double* grad) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    DeviceGuard guard(s->device);
    int rc = ensure_committed(s);
    if (rc) return rc;
    // re-priming over a gradient that may still be uploading: let the stream drain before the pinned
    // staging buffer of this slot is overwritten
    if (s->primed) HIPCHK(hipStreamSynchronize(s->stream));
    rc = stage_grad(s, grad, s->cur);
    if (rc) return rc;
    rc = do_prime(s, s->d_stage[s->cur], s->cur);
    if (rc) return rc;
    s->primed = true;
    s->g_cur = s->d_stage[s->cur];
    s->primed_qindex = -1;
    return 0;
}

int ellhip_cut(ellhip_space* s, int kind, double beta0, int has_beta1, double beta1) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    if (!s->primed) return fail(ELLHIP_E_STATE, "ellhip_cut without a primed gradient");
    if (s->shrink_pending) return fail(ELLHIP_E_STATE, "ellhip_cut: previous cut not committed");
    DeviceGuard guard(s->device);
    CutParams cp;
    int rc = make_params(kind, beta0, has_beta1, beta1, cp);
    if (rc) return rc;
    s->live_arm = true;
    rc = do_cut(s, s->g_cur, nullptr, cp, 0, nullptr, nullptr);
    s->live_arm = false;
    if (rc) s->live_done = false;
    if (!rc) rc = live_publish(s);
    if (!rc) rc = live_wait(s);
    if (rc) return rc;
    const int status = s->h_result->status;
    s->npend = s->h_result->npend;  // deferred mode: a failed cut records nothing
    s->shrink_pending = (status == ELLHIP_SUCCESS) && s->variant == ELLHIP_SPACE_ELL && !deferring(s);
    s->primed = false;  // the gradient has been consumed; gt[cur] stays valid for the pending shrink
    return status;
}

int ellhip_commit(ellhip_space* s, const double* next_grad) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    if (s->in_two_phase) return fail(ELLHIP_E_STATE, "update_begin without update_end");
    DeviceGuard guard(s->device);
    const int nslot = s->cur ^ 1;
    if (next_grad) {
        int rc = stage_grad(s, next_grad, nslot);
        if (rc) return rc;
    }
    const bool shrink = s->shrink_pending;
    int rc = do_commit(s, shrink, next_grad ? s->d_stage[nslot] : nullptr);
    if (rc) return rc;
    if (shrink) s->needs_mirror = false;  // the mirror ran ahead of this successful shrink
    s->shrink_pending = false;
    if (next_grad) {
        s->cur = nslot;
        s->primed = true;
        s->g_cur = s->d_stage[nslot];
        s->primed_qindex = -1;
    } else {
        drop_prime(s);
    }
    return 0;
}

// ---- the SearchSpace update and its two-phase form ------------------------------------------------

int ellhip_update_begin(ellhip_space* s, int kind, const double* grad, double beta0, int has_beta1, double beta1) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    if (s->in_two_phase) return fail(ELLHIP_E_STATE, "update_begin called twice");
    if (!grad) return fail(ELLHIP_E_INVALID, "grad is NULL");
    CutParams cp;
    int rc = make_params(kind, beta0, has_beta1, beta1, cp);
    if (rc) return rc;
    rc = ellhip_prime(s, grad);
    if (rc) return rc;
    s->two_phase_cp = cp;
    s->in_two_phase = true;
    return 0;
}

int ellhip_update_end(ellhip_space* s) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    if (!s->in_two_phase) return fail(ELLHIP_E_STATE, "update_end without update_begin");
    DeviceGuard guard(s->device);
    s->in_two_phase = false;
    s->live_arm = true;
    int rc = do_cut(s, s->g_cur, nullptr, s->two_phase_cp, 0, nullptr, nullptr);
    s->live_arm = false;
    if (rc) {
        s->live_done = false;
        return rc;
    }
    // status, tsq, kappa and the new centre are final after the scalar stage: they go to the host from there
    rc = live_publish(s);
    if (rc) return rc;
    // the rank-1 pass (or, on the recorded schedule, the apply pass when the slots are full) is skipped on the device when
    // the cut failed, so it is issued before the status is known -- and nobody waits for it: the caller's next statements
    // (status test, tsq, xc(), its oracle) run beside it, the next update on this handle is ordered behind it by the stream
    rc = do_commit(s, true, nullptr);
    if (rc) return rc;
    drop_prime(s);
    rc = live_wait(s);
    if (rc) return rc;
    const int status = s->h_result->status;
    if (s->variant == ELLHIP_SPACE_ELL && s->npend > 0 && status != ELLHIP_SUCCESS && s->h_result->npend < s->npend)
        s->npend = s->h_result->npend;  // deferred mode: the failed cut recorded nothing
    if (status == ELLHIP_SUCCESS) s->needs_mirror = false;
    return status;
}

int ellhip_update(ellhip_space* s, int kind, const double* grad, double beta0, int has_beta1, double beta1) {
    int rc = ellhip_update_begin(s, kind, grad, beta0, has_beta1, beta1);
    if (!rc) rc = ellhip_update_end(s);
    return rc;
}

namespace {
// kappa / tsq are cached on the host by every synchronous call; after asynchronous queue cuts they are fetched
void refresh_scalars(const ellhip_space* s_c) {
    if (!s_c->scalars_stale) return;
    ellhip_space* s = const_cast<ellhip_space*>(s_c);
    DeviceGuard guard(s->device);
    (void)read_back(s);
}
}  // namespace

double ellhip_tsq(const ellhip_space* s) {
    if (!s) return 0.0;
    refresh_scalars(s);
    return s->tsq;
}
double ellhip_kappa(const ellhip_space* s) {
    if (!s) return 0.0;
    refresh_scalars(s);
    return s->kappa;
}
int64_t ellhip_ndim(const ellhip_space* s) { return s ? s->n : 0; }

int ellhip_get_xc(const ellhip_space* s, double* xc_out) {
    if (!s || !xc_out) return fail(ELLHIP_E_INVALID, "NULL argument");
    if (s->xc_host_valid) {  // published by the last direct update (k_publish): no device call at all
        memcpy(xc_out, s->h_xc, (size_t)s->n * sizeof(double));
        return 0;
    }
    DeviceGuard guard(s->device);
    HIPCHK(hipMemcpyAsync(xc_out, s->d_xc, (size_t)s->n * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}

int ellhip_set_xc(ellhip_space* s, const double* xc) {
    if (!s || !xc) return fail(ELLHIP_E_INVALID, "NULL argument");
    DeviceGuard guard(s->device);
    // staged through the slot that is NOT holding a primed gradient
    s->xc_host_valid = false;
    const int slot = s->cur ^ 1;
    HIPCHK(hipStreamSynchronize(s->stream));
    memcpy(s->h_stage[slot], xc, (size_t)s->n * sizeof(double));
    HIPCHK(hipMemcpyAsync(s->d_xc, s->h_stage[slot], (size_t)s->n * sizeof(double), hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}

int ellhip_get_mq(const ellhip_space* s_c, double* mq_out) {
    if (!s_c || !mq_out) return fail(ELLHIP_E_INVALID, "NULL argument");
    ellhip_space* s = const_cast<ellhip_space*>(s_c);
    DeviceGuard guard(s->device);
    int rc = make_q_current(s);
    if (rc) return rc;
    HIPCHK(hipMemcpy2DAsync(mq_out, (size_t)s->n * sizeof(double), s->d_Q, (size_t)s->ld * sizeof(double),
                            (size_t)s->n * sizeof(double), (size_t)s->nrows, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}

int ellhip_set_no_defer_trick(ellhip_space* s, int flag) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    if (s->variant != ELLHIP_SPACE_ELL) return fail(ELLHIP_E_INVALID, "no_defer_trick exists on Ell only");
    if (s->shard_symmetric && flag) return fail(ELLHIP_E_STATE, "symmetric row shard: no_defer_trick is not available");
    DeviceGuard guard(s->device);
    int rc = make_q_current(s);  // recorded updates belong to the old mode
    if (rc) return rc;
    s->no_defer_trick = flag ? 1 : 0;
    return 0;
}

int ellhip_set_defer_depth(ellhip_space* s, int depth) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    if (s->variant != ELLHIP_SPACE_ELL) return fail(ELLHIP_E_INVALID, "deferred shrink exists on Ell only");
    if (depth != 1 && depth != 8 && depth != 16 && depth != 24) return fail(ELLHIP_E_INVALID, "defer depth must be 1, 8, 16 or 24");
    if (s->shard_symmetric && depth == 1 && s->upper_stale)
        return fail(ELLHIP_E_STATE, "symmetric row shard: the rows are current up to their diagonal only");
    DeviceGuard guard(s->device);
    int rc = make_q_current(s);
    if (rc) return rc;
    if (depth > 1) {
        rc = symv_alloc(s);
        if (rc) return rc;
    }
    if (depth >= 16) {
        // 16 / 24 pending updates only fit the lower-triangle schedule (k_symv + k_apply_lower)
        const bool lower_schedule = s->symv && s->apply_lower && (s->n % 2) == 0 && s->d_rowpart &&
                                    (s->sharded ? s->shard_symmetric : s->n >= s->symv_min_n);
        if (!lower_schedule)
            return fail(ELLHIP_E_INVALID, "defer depth 16 / 24 needs the lower-triangle schedule: an unsharded handle with n "
                                          "even and >= ELLHIP_OPT_SYMV_MIN_N (5120 by default), or a symmetric row shard");
    }
    s->defer = depth;
    return 0;
}

int ellhip_defer_depth(const ellhip_space* s) { return s ? s->defer : 0; }

int ellhip_flush(ellhip_space* s) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    if (s->variant != ELLHIP_SPACE_ELL) return 0;
    DeviceGuard guard(s->device);
    const bool uncut = prime_is_uncut(s);
    int rc = ensure_committed(s);
    if (rc) return rc;
    if (s->npend > 0) {
        rc = flush_pending(s, nullptr, nullptr);
        if (rc) return rc;
        if (uncut) return refresh_prime(s);  // the primed gradient's Q_base*g belongs to the old base
    }
    return 0;
}

int64_t ellhip_queue_primed(const ellhip_space* s) { return (s && s->primed) ? (int64_t)s->primed_qindex : -1; }

int ellhip_set_shard_symmetric(ellhip_space* s, int flag) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    if (!s->sharded || s->variant != ELLHIP_SPACE_ELL) return fail(ELLHIP_E_INVALID, "symmetric mode is for row shards of Ell");
    if ((s->n % 2) != 0 || (s->row0 % SYMV_H) != 0 || ((s->row0 + s->nrows) % SYMV_H != 0 && s->row0 + s->nrows != s->n))
        return fail(ELLHIP_E_INVALID, "symmetric row shard: n must be even and the shard boundaries multiples of 64");
    if (s->in_two_phase || s->primed || s->shrink_pending || s->npend > 0 || s->upper_stale)
        return fail(ELLHIP_E_STATE, "symmetric row shard: set the mode before the first update");
    DeviceGuard guard(s->device);
    s->shard_symmetric = flag != 0;
    if (s->shard_symmetric && s->defer > 1) {
        int rc = symv_alloc(s);
        if (rc) return rc;
    }
    return 0;
}

// ---- options ----------------------------------------------------------------------------------
namespace {
int option_ok(int key, long long v) {
    switch (key) {
        case ELLHIP_OPT_APPLY_KERNEL:
            return (v >= -1 && v <= 2) ? 0 : fail(ELLHIP_E_INVALID, "option value must be -1, 0, 1 or 2");
        case ELLHIP_OPT_STAGE_DIRECT: return (v == 0 || v == 1) ? 0 : fail(ELLHIP_E_INVALID, "option value must be 0 or 1");
        case ELLHIP_OPT_AUTO_DEFER: case ELLHIP_OPT_SYMV: case ELLHIP_OPT_APPLY_LOWER:
        case ELLHIP_OPT_FUSE_DOTS:
            return (v == 0 || v == 1) ? 0 : fail(ELLHIP_E_INVALID, "option value must be 0 or 1");
        case ELLHIP_OPT_RESIDENT: return (v >= 0 && v <= 2) ? 0 : fail(ELLHIP_E_INVALID, "ELLHIP_OPT_RESIDENT: 0, 1 or 2");
        case ELLHIP_OPT_OVERLAP: return (v >= 0 && v <= 2) ? 0 : fail(ELLHIP_E_INVALID, "ELLHIP_OPT_OVERLAP: 0, 1 or 2");
        case ELLHIP_OPT_LOOKAHEAD: return (v >= 1 && v <= 32) ? 0 : fail(ELLHIP_E_INVALID, "ELLHIP_OPT_LOOKAHEAD: 1 .. 32");
        case ELLHIP_OPT_QUEUE_DEPTH: return (v == 0 || v == 48) ? 0 : fail(ELLHIP_E_INVALID, "ELLHIP_OPT_QUEUE_DEPTH: 0 or 48");
        case ELLHIP_OPT_RESIDENT_FAULT: return v >= -1 ? 0 : fail(ELLHIP_E_INVALID, "ELLHIP_OPT_RESIDENT_FAULT: -1 or a cut index");
        case ELLHIP_OPT_SYMV_MIN_N: return v >= 512 ? 0 : fail(ELLHIP_E_INVALID, "ELLHIP_OPT_SYMV_MIN_N must be >= 512");
        case ELLHIP_OPT_STABLE_SOLVE: return (v >= 0 && v <= 3) ? 0 : fail(ELLHIP_E_INVALID, "ELLHIP_OPT_STABLE_SOLVE: 0 .. 3");
        case ELLHIP_OPT_STABLE_FACTOR:
            return (v >= 0 && v <= 2) ? 0 : fail(ELLHIP_E_INVALID, "option value must be 0, 1 or 2");
        case ELLHIP_OPT_PAD: return (v >= -1 && v <= 4096) ? 0 : fail(ELLHIP_E_INVALID, "ELLHIP_OPT_PAD: -1 .. 4096");
        case ELLHIP_OPT_LP_GRID: return (v >= 0 && v <= 65536) ? 0 : fail(ELLHIP_E_INVALID, "ELLHIP_OPT_LP_GRID: 0 .. 65536");
        case ELLHIP_OPT_LP_WIDE: return (v >= -1 && v <= 1) ? 0 : fail(ELLHIP_E_INVALID, "ELLHIP_OPT_LP_WIDE: -1, 0, 1");
        case ELLHIP_OPT_BATCH_THREADS:
            return (v == 0 || v == 64 || v == 128 || v == 256) ? 0 : fail(ELLHIP_E_INVALID, "ELLHIP_OPT_BATCH_THREADS: 0, 64, 128, 256");
        default: return fail(ELLHIP_E_INVALID, "unknown option key");
    }
}
}  // namespace

int ellhip_set_default_option(int key, int64_t value) {
    int rc = option_ok(key, value);
    if (rc) return rc;
    switch (key) {
        case ELLHIP_OPT_AUTO_DEFER: g_defaults.auto_defer = (int)value; break;
        case ELLHIP_OPT_SYMV: g_defaults.symv = (int)value; break;
        case ELLHIP_OPT_SYMV_MIN_N: g_defaults.symv_min_n = value; break;
        case ELLHIP_OPT_APPLY_LOWER: g_defaults.apply_lower = (int)value; break;
        case ELLHIP_OPT_APPLY_KERNEL: g_defaults.apply_kernel = (int)value; break;
        case ELLHIP_OPT_FUSE_DOTS: g_defaults.fuse_dots = (int)value; break;
        case ELLHIP_OPT_RESIDENT: g_defaults.resident = (int)value; break;
        case ELLHIP_OPT_OVERLAP: g_defaults.overlap = (int)value; break;
        case ELLHIP_OPT_LOOKAHEAD: g_defaults.lookahead = (int)value; break;
        case ELLHIP_OPT_QUEUE_DEPTH: g_defaults.queue_depth = (int)value; break;
        case ELLHIP_OPT_STABLE_SOLVE: g_defaults.stable_solve = (int)value; break;
        case ELLHIP_OPT_STABLE_FACTOR: g_defaults.stable_factor = (int)value; break;
        case ELLHIP_OPT_PAD: g_defaults.pad = (int)value; break;
        case ELLHIP_OPT_LP_GRID: g_defaults.lp_grid = (int)value; break;
        case ELLHIP_OPT_LP_WIDE: g_defaults.lp_wide = (int)value; break;
        case ELLHIP_OPT_BATCH_THREADS: g_defaults.batch_threads = (int)value; break;
        case ELLHIP_OPT_STAGE_DIRECT: g_defaults.stage_direct = (int)value; break;
    }
    return 0;
}

int ellhip_default_option(int key, int64_t* value) {
    if (!value) return fail(ELLHIP_E_INVALID, "value is NULL");
    switch (key) {
        case ELLHIP_OPT_AUTO_DEFER: *value = g_defaults.auto_defer; break;
        case ELLHIP_OPT_SYMV: *value = g_defaults.symv; break;
        case ELLHIP_OPT_SYMV_MIN_N: *value = g_defaults.symv_min_n; break;
        case ELLHIP_OPT_APPLY_LOWER: *value = g_defaults.apply_lower; break;
        case ELLHIP_OPT_APPLY_KERNEL: *value = g_defaults.apply_kernel; break;
        case ELLHIP_OPT_FUSE_DOTS: *value = g_defaults.fuse_dots; break;
        case ELLHIP_OPT_RESIDENT: *value = g_defaults.resident; break;
        case ELLHIP_OPT_OVERLAP: *value = g_defaults.overlap; break;
        case ELLHIP_OPT_LOOKAHEAD: *value = g_defaults.lookahead; break;
        case ELLHIP_OPT_QUEUE_DEPTH: *value = g_defaults.queue_depth; break;
        case ELLHIP_OPT_STABLE_SOLVE: *value = g_defaults.stable_solve; break;
        case ELLHIP_OPT_STABLE_FACTOR: *value = g_defaults.stable_factor; break;
        case ELLHIP_OPT_PAD: *value = g_defaults.pad; break;
        case ELLHIP_OPT_LP_GRID: *value = g_defaults.lp_grid; break;
        case ELLHIP_OPT_LP_WIDE: *value = g_defaults.lp_wide; break;
        case ELLHIP_OPT_BATCH_THREADS: *value = g_defaults.batch_threads; break;
        case ELLHIP_OPT_STAGE_DIRECT: *value = g_defaults.stage_direct; break;
        default: return fail(ELLHIP_E_INVALID, "unknown option key");
    }
    return 0;
}

int ellhip_set_option(ellhip_space* s, int key, int64_t value) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    int rc = option_ok(key, value);
    if (rc) return rc;
    DeviceGuard guard(s->device);
    const bool ell = s->variant == ELLHIP_SPACE_ELL;
    switch (key) {
        // chosen per queue run: nothing recorded depends on them
        case ELLHIP_OPT_RESIDENT:
            if (!ell) return fail(ELLHIP_E_INVALID, "this option exists on Ell only");
            s->resident = (int)value;
            return 0;
        case ELLHIP_OPT_OVERLAP:
            if (!ell) return fail(ELLHIP_E_INVALID, "this option exists on Ell only");
            s->overlap = (int)value;
            return 0;
        case ELLHIP_OPT_LOOKAHEAD:
            if (!ell) return fail(ELLHIP_E_INVALID, "this option exists on Ell only");
            s->lookahead = (int)value;
            return 0;
        case ELLHIP_OPT_QUEUE_DEPTH:
            if (!ell) return fail(ELLHIP_E_INVALID, "this option exists on Ell only");
            s->queue_depth = (int)value;
            return 0;
        case ELLHIP_OPT_RESIDENT_FAULT:
            if (!ell) return fail(ELLHIP_E_INVALID, "this option exists on Ell only");
            s->rs_fault_at = value;
            return 0;
        // what has been recorded (and a stale upper triangle) belongs to the schedule in force: Q is made current first
        case ELLHIP_OPT_SYMV: case ELLHIP_OPT_SYMV_MIN_N: case ELLHIP_OPT_APPLY_LOWER: case ELLHIP_OPT_APPLY_KERNEL:
        case ELLHIP_OPT_FUSE_DOTS: {
            if (!ell) return fail(ELLHIP_E_INVALID, "this option exists on Ell only");
            if (s->in_two_phase) return fail(ELLHIP_E_STATE, "update_begin without update_end");
            rc = make_q_current(s);
            if (rc) return rc;
            if (key == ELLHIP_OPT_SYMV) s->symv = (int)value;
            else if (key == ELLHIP_OPT_SYMV_MIN_N) s->symv_min_n = value;
            else if (key == ELLHIP_OPT_APPLY_LOWER) s->apply_lower = (int)value;
            else if (key == ELLHIP_OPT_APPLY_KERNEL) s->apply_kernel = (int)value;
            else s->fuse_dots = (int)value;
            if (s->defer > 1) {
                rc = symv_alloc(s);  // (a lowered threshold may bring the lower-triangle schedule into reach)
                if (rc) return rc;
            }
            if (s->defer >= 16) {  // depth 16 / 24 exist on the lower-triangle schedule only: fall back to 8 where it is gone
                const bool lower = s->symv && s->apply_lower && (s->n % 2) == 0 && s->d_rowpart &&
                                   (s->sharded ? s->shard_symmetric : s->n >= s->symv_min_n);
                if (!lower) s->defer = 8;
            }
            return 0;
        }
        case ELLHIP_OPT_STABLE_SOLVE: case ELLHIP_OPT_STABLE_FACTOR:
            if (ell) return fail(ELLHIP_E_INVALID, "this option exists on EllStable only");
            rc = stable_mirror_leave(s);  // (the next update enters the layout its options ask for)
            if (rc) return rc;
            HIPCHK(hipStreamSynchronize(s->stream));
            if (key == ELLHIP_OPT_STABLE_SOLVE) s->stable_solve = (int)value; else s->stable_factor = (int)value;
            return 0;
        default:
            return fail(ELLHIP_E_INVALID, "this option is a creation-time default only (ellhip_set_default_option)");
    }
}

int ellhip_get_option(const ellhip_space* s, int key, int64_t* value) {
    if (!s || !value) return fail(ELLHIP_E_INVALID, "NULL argument");
    switch (key) {
        case ELLHIP_OPT_SYMV: *value = s->symv; break;
        case ELLHIP_OPT_SYMV_MIN_N: *value = s->symv_min_n; break;
        case ELLHIP_OPT_APPLY_LOWER: *value = s->apply_lower; break;
        case ELLHIP_OPT_APPLY_KERNEL: *value = s->apply_kernel; break;
        case ELLHIP_OPT_FUSE_DOTS: *value = s->fuse_dots; break;
        case ELLHIP_OPT_RESIDENT: *value = s->resident; break;
        case ELLHIP_OPT_OVERLAP: *value = s->overlap; break;
        case ELLHIP_OPT_LOOKAHEAD: *value = s->lookahead; break;
        case ELLHIP_OPT_QUEUE_DEPTH: *value = s->queue_depth; break;
        case ELLHIP_OPT_RESIDENT_FAULT: *value = s->rs_fault_at; break;
        case ELLHIP_OPT_RESIDENT_ABANDONED: *value = s->rs_abandoned; break;
        case ELLHIP_OPT_STABLE_SOLVE: *value = s->stable_solve; break;
        case ELLHIP_OPT_STABLE_FACTOR: *value = s->stable_factor; break;
        case ELLHIP_OPT_STABLE_MIRRORED: *value = s->st_mirrored ? 1 : 0; break;
        case ELLHIP_OPT_STAGE_DIRECT: *value = s->stage_direct ? 1 : 0; break;
        case ELLHIP_OPT_PAD: *value = s->ld - s->n; break;
        default: return fail(ELLHIP_E_INVALID, "not a per-handle option");
    }
    return 0;
}

int ellhip_set_use_parallel_cut(ellhip_space* s, int flag) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    s->use_parallel_cut = flag ? 1 : 0;
    return 0;
}

int ellhip_calc(int64_t n, int use_parallel_cut, int kind, double beta0, int has_beta1, double beta1, double tsq,
                double* out3, int device) {
    if (!out3 || n < 1 || kind < 0 || kind > 2) return fail(ELLHIP_E_INVALID, "bad argument");
    if (ellhip_device_count() <= 0) return fail(ELLHIP_E_NODEVICE, "no HIP device");
    if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
    DeviceGuard guard(device);
    double* d_out = nullptr;
    HIPCHK(hipMalloc(&d_out, 4 * sizeof(double)));
    CutParams cp;
    cp.kind = kind;
    cp.has_b1 = has_beta1 ? 1 : 0;
    cp.b0 = beta0;
    cp.b1 = has_beta1 ? beta1 : 0.0;
    hipLaunchKernelGGL(k_calc_one, dim3(1), dim3(1), 0, 0, EllCalcDev::make(n, use_parallel_cut), cp, tsq, d_out);
    double h[4];
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(ELLHIP_E_HIP, "ellhip_calc", e);
    out3[0] = h[1];
    out3[1] = h[2];
    out3[2] = h[3];
    return (int)h[0];
}

double* ellhip_gt_dev(ellhip_space* s) { return s ? s->d_gt[s->cur] : nullptr; }

int ellhip_set_gt_dev(ellhip_space* s, double* gt_dev_a, double* gt_dev_b) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    if (s->in_two_phase || s->primed || s->shrink_pending) return fail(ELLHIP_E_STATE, "update in flight");
    s->d_gt[0] = gt_dev_a ? gt_dev_a : s->d_gt_own[0];
    s->d_gt[1] = gt_dev_b ? gt_dev_b : s->d_gt_own[1];
    return 0;
}

// ---- queue ------------------------------------------------------------------------------------

int ellhip_queue_upload(ellhip_space* s, int64_t k, const int32_t* kinds, const double* grads, const double* beta0,
                        const int32_t* has_beta1, const double* beta1) {
    if (!s || k < 1 || !kinds || !grads || !beta0) return fail(ELLHIP_E_INVALID, "bad argument");
    DeviceGuard guard(s->device);
    int rc = ensure_committed(s);
    if (rc) return rc;
    drop_prime(s);
    HIPCHK(hipStreamSynchronize(s->stream));
    queue_free(s);
    const long long n = s->n;
    std::vector<CutParams> hp((size_t)k);
    for (int64_t i = 0; i < k; ++i) {
        if (kinds[i] < 0 || kinds[i] > 2) return fail(ELLHIP_E_INVALID, "bad cut kind in queue");
        hp[(size_t)i].kind = kinds[i];
        hp[(size_t)i].has_b1 = (has_beta1 && has_beta1[i]) ? 1 : 0;
        hp[(size_t)i].b0 = beta0[i];
        hp[(size_t)i].b1 = (has_beta1 && has_beta1[i] && beta1) ? beta1[i] : 0.0;
    }
    HIPCHK(hipMalloc(&s->d_qparams, (size_t)k * sizeof(CutParams)));
    HIPCHK(hipMalloc(&s->d_qgrads, (size_t)k * (size_t)n * sizeof(double)));
    HIPCHK(hipMalloc(&s->d_qstatus, (size_t)k * sizeof(int)));
    HIPCHK(hipMalloc(&s->d_qtsq, (size_t)k * sizeof(double)));
    HIPCHK(hipMemcpy(s->d_qparams, hp.data(), (size_t)k * sizeof(CutParams), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->d_qgrads, grads, (size_t)k * (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(fill_now(s->d_qstatus, 0xff, (size_t)k * sizeof(int), s->stream));  // -1 = not run yet
    HIPCHK(fill_now(s->d_qtsq, 0, (size_t)k * sizeof(double), s->stream));
    s->qk = k;
    return 0;
}

int ellhip_queue_prime(ellhip_space* s, int64_t index) {
    int rc = queue_index_ok(s, index);
    if (rc) return rc;
    DeviceGuard guard(s->device);
    return queue_prime_impl(s, index);
}

int ellhip_queue_cut(ellhip_space* s, int64_t index) {
    int rc = queue_index_ok(s, index);
    if (rc) return rc;
    DeviceGuard guard(s->device);
    return queue_cut_impl(s, index);
}

int ellhip_queue_commit(ellhip_space* s, int64_t index, int64_t next_index) {
    int rc = queue_index_ok(s, index);
    if (rc) return rc;
    if (next_index >= s->qk) return fail(ELLHIP_E_INVALID, "queue next_index out of range");
    DeviceGuard guard(s->device);
    return queue_commit_impl(s, index, next_index);
}

int ellhip_queue_begin(ellhip_space* s, int64_t index) { return ellhip_queue_prime(s, index); }

int ellhip_queue_end(ellhip_space* s, int64_t index) {
    int rc = ellhip_queue_cut(s, index);
    if (rc) return rc;
    return ellhip_queue_commit(s, index, -1);
}

int ellhip_queue_run(ellhip_space* s, int64_t first, int64_t count) {
    if (!s || first < 0 || count < 0 || first + count > s->qk) return fail(ELLHIP_E_INVALID, "queue range");
    DeviceGuard guard(s->device);
    if (resident_ok(s, count)) {
        const int rrc = resident_run(s, first, count);
        if (rrc != RS_FALLBACK) return rrc;
    }
    for (int64_t i = first; i < first + count; ++i) {
        int rc = queue_prime_impl(s, i);
        if (!rc) rc = queue_cut_impl(s, i);
        if (!rc) rc = queue_commit_impl(s, i, -1);
        if (rc) return rc;
    }
    return 0;
}

int ellhip_queue_run_fused(ellhip_space* s, int64_t first, int64_t count) {
    if (!s || first < 0 || count < 0 || first + count > s->qk) return fail(ELLHIP_E_INVALID, "queue range");
    DeviceGuard guard(s->device);
    if (resident_ok(s, count)) {
        const int rrc = resident_run(s, first, count);
        if (rrc != RS_FALLBACK) return rrc;
    }
    if (multi_ok(s)) {
        const int mrc = queue_run_multi(s, first, count);
        if (mrc != MULTI_NO_MEMORY) return mrc;
    }
    if (overlap_ok(s)) return queue_run_overlapped(s, first, count);
    for (int64_t i = first; i < first + count; ++i) {
        int rc = queue_prime_impl(s, i);  // (pipelined form: only the first cut of a run pays a separate GEMV pass)
        if (!rc) rc = queue_cut_impl(s, i);
        if (!rc) rc = queue_commit_impl(s, i, (i + 1 < s->qk) ? i + 1 : -1);
        if (rc) return rc;
    }
    return 0;
}

int ellhip_queue_results(ellhip_space* s, int32_t* status_out, double* tsq_out) {
    if (!s || s->qk < 1) return fail(ELLHIP_E_INVALID, "no queue");
    DeviceGuard guard(s->device);
    int rc = read_back(s);
    if (rc) return rc;
    if (s->variant == ELLHIP_SPACE_ELL && s->h_result->halted) s->npend = s->h_result->npend;
    std::vector<int32_t> st((size_t)s->qk);
    HIPCHK(hipMemcpy(st.data(), s->d_qstatus, (size_t)s->qk * sizeof(int), hipMemcpyDeviceToHost));
    if (status_out) memcpy(status_out, st.data(), (size_t)s->qk * sizeof(int));
    if (tsq_out) HIPCHK(hipMemcpy(tsq_out, s->d_qtsq, (size_t)s->qk * sizeof(double), hipMemcpyDeviceToHost));
    if (s->needs_mirror) {
        for (int64_t i = 0; i < s->qk; ++i)
            if (st[(size_t)i] == ELLHIP_SUCCESS) {
                s->needs_mirror = false;
                break;
            }
    }
    // a halted queue stays halted until its results have been read; then direct updates work again
    if (s->h_result->halted && s->variant == ELLHIP_SPACE_ELL_STABLE) {
        // (mirrored layout: a k_st_mirror_enter issued while the queue was halted did nothing; host and device agree again)
        rc = stable_mirror_leave(s);
        if (rc) return rc;
    }
    if (s->h_result->halted) {
        s->h_result->halted = 0;
        s->h_result->stop = STOP_NONE;
        HIPCHK(hipMemcpyAsync(s->d_st, s->h_result, sizeof(DevState), hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        drop_prime(s);  // whatever was primed beyond the failing cut never ran
        s->shrink_pending = false;
    }
    return 0;
}

// ---- streams / timing -------------------------------------------------------------------------

int ellhip_set_stream(ellhip_space* s, void* hip_stream) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    DeviceGuard guard(s->device);
    HIPCHK(hipStreamSynchronize(s->stream));
    s->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : s->own_stream;
    return 0;
}

int ellhip_synchronize(ellhip_space* s) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    DeviceGuard guard(s->device);
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}

int ellhip_profile_enable(ellhip_space* s, int flag) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    s->profile = flag != 0;
    return 0;
}

int ellhip_profile_read(ellhip_space* s, double* ms_out, int64_t* count_out) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    DeviceGuard guard(s->device);
    int rc = prof_flush(s);
    if (rc) return rc;
    for (int i = 0; i < ELLHIP_NKERNEL_CLASSES; ++i) {
        if (ms_out) ms_out[i] = s->prof_ms[i];
        if (count_out) count_out[i] = s->prof_cnt[i];
        s->prof_ms[i] = 0.0;
        s->prof_cnt[i] = 0;
    }
    return 0;
}

}  // extern "C"

#include "lowpass_capi.inc.hpp"
#include "batch_capi.inc.hpp"
#include "lmi_capi.inc.hpp"
#include "sharded_capi.inc.hpp"
