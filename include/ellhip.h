/*
 * ellhip.h -- C ABI of the MI355X ellipsoid-update engine (libellhip.so).
 *
 * This is the drop-in boundary for ONE path of luk036/ellalgo-rs 0.1.7: the search-space update
 * behind `trait SearchSpace` (src/cutting_plane.rs:154-182), i.e. Ell::update_core
 * (src/ell.rs:97-137) and EllStable::update_core (src/ell_stable.rs:52-125) together with the
 * EllCalc coefficient stage (src/ell_calc.rs:627-931).  The reference has no FFI today (it is
 * safe single-threaded Rust); these entry points are what a Rust `impl SearchSpace for EllHip`
 * binds (see INTEGRATION.md for the `extern "C"` block and the impl).
 *
 * Conventions
 *   - Plain pointers and sizes only.  Every `const double*` / `double*` argument is a HOST buffer
 *     owned by the caller unless its name ends in `_dev`; the library copies in/out and keeps no
 *     reference after the call returns.  Device memory is owned by the handle.
 *   - Matrices are n*n row-major f64, exactly `Arr{data, rows, cols}` (src/arr.rs:12-16).
 *   - Return value of the update calls: 0..3 = CutStatus in the reference's declaration order
 *     (src/cutting_plane.rs:31-37); negative = library failure (ELLHIP_E_*), which a binding maps
 *     to CutStatus::Unknown.  All other int-returning calls: 0 = ok, negative = ELLHIP_E_*.
 *   - State contract on a non-Success cut (src/ell.rs:105-109): tsq is updated, Q / xc / kappa
 *     are untouched (EllStable additionally has its scratch triangle rewritten,
 *     src/ell_stable.rs:66).
 *   - One handle is used from one host thread at a time (`&mut self` in the reference).  All work
 *     of a handle is issued on one HIP stream (ellhip_set_stream; default: a stream the handle
 *     owns).  ellhip_update / ellhip_update_end are synchronous at return, like the reference.
 *   - There is NO CPU fallback: with no HIP device every call that needs one fails with
 *     ELLHIP_E_NODEVICE.
 */
#ifndef ELLHIP_H
#define ELLHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* CutStatus, src/cutting_plane.rs:31-37 */
#define ELLHIP_SUCCESS 0
#define ELLHIP_NOSOLN 1
#define ELLHIP_NOEFFECT 2
#define ELLHIP_UNKNOWN 3

/* library failures */
#define ELLHIP_E_INVALID (-1)   /* bad argument (NULL, n mismatch, ...); replaces Arr's shape assert!s (src/arr.rs:427-429) */
#define ELLHIP_E_HIP (-2)       /* a HIP runtime call failed; see ellhip_last_error() */
#define ELLHIP_E_NODEVICE (-3)  /* no usable HIP device */
#define ELLHIP_E_NOMEM (-4)
#define ELLHIP_E_STATE (-5)     /* call sequence violated (e.g. update_end without update_begin) */

/* which SearchSpace method is being served (src/cutting_plane.rs:161-179) */
#define ELLHIP_CUT_BIAS 0     /* update_bias_cut    -> CutType::call_bias_cut    (src/ell.rs:189-191,201-203) */
#define ELLHIP_CUT_CENTRAL 1  /* update_central_cut -> CutType::call_central_cut (src/ell.rs:192-194,204-206) */
#define ELLHIP_CUT_Q 2        /* update_q           -> CutType::call_q_cut       (src/ell.rs:195-197,207-209) */

/* search-space variant */
#define ELLHIP_SPACE_ELL 0         /* struct Ell,       src/ell.rs:9-16 */
#define ELLHIP_SPACE_ELL_STABLE 1  /* struct EllStable, src/ell_stable.rs:9-15 */

typedef struct ellhip_space ellhip_space; /* opaque */

/* ---- lifetime -------------------------------------------------------------------------------
 * ellhip_create replaces Ell::new_with_matrix / new / new_with_scalar / from_covariance
 * (src/ell.rs:31-78) and the EllStable constructors (src/ell_stable.rs:18-35):
 *   mq != NULL            : copy of the n*n matrix            (new_with_matrix, from_covariance)
 *   mq == NULL, diag != 0 : diag(diag[0..n))                  (new:             kappa = 1)
 *   both NULL             : identity                          (new_with_scalar: kappa = val)
 * xc == NULL means the zero vector.  device < 0 = current device.
 * Ell requires the matrix it is given to be symmetric only in the sense the reference does: after
 * the first successful update the upper triangle is a copy of the lower one (src/ell.rs:124-126);
 * the engine reproduces exactly that (it reads Q[c][r] for c > r on the first update of a matrix
 * that was supplied by the caller and never again).
 */
int ellhip_create(ellhip_space **out, int variant, int64_t n, double kappa, const double *mq,
                  const double *diag, const double *xc, int device);

/* Row-block shard of an n*n Ell for the multi-GPU schedule: this handle owns rows
 * [row0, row0 + nrows) of Q (mq, if given, is the nrows*n block).  Vectors are full length. */
int ellhip_create_shard(ellhip_space **out, int64_t n, int64_t row0, int64_t nrows, double kappa,
                        const double *mq_rows, const double *diag, const double *xc, int device);

/* `#[derive(Clone)]` on Ell / EllStable (src/ell.rs:8, src/ell_stable.rs:8), used by
 * BSearchAdaptor::assess_bs (src/cutting_plane.rs:410).  Device-to-device copy. */
int ellhip_clone(const ellhip_space *src, ellhip_space **out);
/* Rust Drop. */
void ellhip_destroy(ellhip_space *s);

/* ---- the SearchSpace methods ---------------------------------------------------------------- */

/* update_bias_cut / update_central_cut / update_q (src/ell.rs:153-175, src/ell_stable.rs:139-161).
 * SingleCut(b) -> beta0 = b, has_beta1 = 0.  ParallelCut(b0, Some(b1)) -> has_beta1 = 1.
 * ParallelCut(b0, None) -> has_beta1 = 0 (same arithmetic, src/ell_calc.rs:677-681).
 * grad: n doubles.  Returns CutStatus (0..3) or ELLHIP_E_*. */
int ellhip_update(ellhip_space *s, int kind, const double *grad, double beta0, int has_beta1,
                  double beta1);
/* SearchSpace::tsq (src/ell.rs:149-151). */
double ellhip_tsq(const ellhip_space *s);
/* SearchSpace::xc (src/ell.rs:144-146): owned copy out, n doubles. */
int ellhip_get_xc(const ellhip_space *s, double *xc_out);
/* SearchSpace::set_xc (src/ell.rs:177-179). */
int ellhip_set_xc(ellhip_space *s, const double *xc);

/* ---- public fields of Ell (src/ell.rs:10-15) and parity observability ----------------------- */
double ellhip_kappa(const ellhip_space *s);
int64_t ellhip_ndim(const ellhip_space *s);
/* Local row block of the matrix (whole n*n for an unsharded handle), row-major. */
int ellhip_get_mq(const ellhip_space *s, double *mq_out);
/* Ell.no_defer_trick (src/ell.rs:10,132-135).  Ell only. */
int ellhip_set_no_defer_trick(ellhip_space *s, int flag);
/* EllCalc.use_parallel_cut (src/ell_calc.rs:630); the reference fixes it to true for Ell. */
int ellhip_set_use_parallel_cut(ellhip_space *s, int flag);

/* ---- EllCalc on the device (src/ell_calc.rs:671-931) ----------------------------------------
 * Runs the same device routine the update kernels use on one lane and returns
 * (status, rho, sigma, delta) for dimension n; out3 = {rho, sigma, delta}. */
int ellhip_calc(int64_t n, int use_parallel_cut, int kind, double beta0, int has_beta1,
                double beta1, double tsq, double *out3, int device);

/* ---- two-phase update for the row-partitioned (multi-GPU) schedule --------------------------
 * begin: uploads grad, runs the local GEMV gt[row0..row0+nrows) = Q_rows * grad into the handle's
 *        full-length gt buffer (device), asynchronously on the handle's stream.
 * The caller then assembles the full gt across ranks IN PLACE on that buffer (RCCL all-gather of
 * nrows doubles per rank; ellhip_gt_dev gives the pointer) on the same stream or one ordered
 * after it.
 * end:   every rank redundantly runs the scalar stage (omega, tsq, EllCalc, xc, kappa) and then the
 *        rank-1 update of its own rows; synchronous at return; returns CutStatus. */
int ellhip_update_begin(ellhip_space *s, int kind, const double *grad, double beta0,
                        int has_beta1, double beta1);
int ellhip_update_end(ellhip_space *s);
/* Device pointer of the full-length (n doubles) gt buffer of the most recently primed gradient
 * (the one to all-gather). */
double *ellhip_gt_dev(ellhip_space *s);
/* Use caller-provided device memory (n doubles each) as the two gt buffers, e.g. tensors a
 * collective library already knows (the pipelined schedule alternates between them; the two-phase
 * schedule only uses the one ellhip_gt_dev returns).  NULLs restore the handle's own buffers. */
int ellhip_set_gt_dev(ellhip_space *s, double *gt_dev_a, double *gt_dev_b);

/* ---- pipelined update: 16 n^2 instead of 24 n^2 bytes per update (Ell) -----------------------
 * Ell::update_core reads Q twice: once for gt = Q*g (src/ell.rs:102) and once for the rank-1 shrink
 * (src/ell.rs:117-128).  The NEXT cut's gradient only depends on the new centre, which the scalar
 * stage (src/ell.rs:103-115) produces BEFORE the shrink, so the shrink of update k and the GEMV of
 * update k+1 can share one pass over Q.  Results are bit-identical to ellhip_update.
 *   ellhip_prime(grad)      gt = Q*grad for the first cut                          (asynchronous)
 *   ellhip_cut(kind, betas) scalar stage of the primed cut: status, tsq, xc, kappa are final and
 *                           observable at return (synchronous); Q is not shrunk yet
 *   ellhip_commit(next)     shrink Q for the cut just taken (if it succeeded) and, in the same pass,
 *                           prime `next` (NULL: shrink only)                       (asynchronous)
 * Driver shape (equivalent to src/cutting_plane.rs:299-311):
 *   prime(g0); loop { st = cut(..); if stop { commit(NULL); break; } x = get_xc(); (g, beta) =
 *   oracle(x); commit(g); }
 * Any other call that needs Q (update, get_mq, clone, ...) commits a pending shrink first.
 * EllStable accepts the same calls but gains nothing (its work happens in ellhip_cut). */
int ellhip_prime(ellhip_space *s, const double *grad);
int ellhip_cut(ellhip_space *s, int kind, double beta0, int has_beta1, double beta1);
int ellhip_commit(ellhip_space *s, const double *next_grad);

/* ---- deferred shrink: 8 n^2 (1 + 1/8) bytes per update (Ell) -----------------------------------
 * depth = 1: Q is rewritten at every successful cut, exactly as src/ell.rs:117-128 does.  This is what a new handle
 * starts with, EXCEPT an unsharded Ell handle with n >= 3072: it starts at depth 24 when n is even and >= 5120 (ELLHIP_OPT_SYMV_MIN_N) and
 * at depth 8 otherwise (the fastest schedules at those sizes; same results to the parity tolerance).
 * ellhip_set_default_option(ELLHIP_OPT_AUTO_DEFER, 0) makes every later handle start at depth 1;
 * ellhip_set_defer_depth(h, 1) does it for one handle.
 * depth = 8: successful cuts are RECORDED as pairs (sigma/omega, gt); the next GEMV reads the unchanged
 * matrix (a read-only pass) and is corrected with the recorded pairs,
 *     gt = Q_base*g - sum_j c_j (v_j.g) v_j ,   omega = g.(Q_base*g) - sum_j c_j (v_j.g)^2 ,
 * and every 8th cut one pass applies the 8 recorded updates element by element in order, i.e. with the
 * same sequence of roundings per element as the reference loop.  xc, kappa, tsq and status are always
 * current; Q is made current before ellhip_get_mq / ellhip_clone return.  Results stay within the
 * 1e-10 parity tolerance of depth 1 (gt is the same vector computed in a different order); they are
 * bit-identical across schedules, row partitions and GPU counts for a given depth.  Not used while
 * no_defer_trick is set or before a non-symmetric input matrix has been mirrored.  Ell only.
 * On an unsharded handle (even n >= 5120) depth 8 also computes Q_base*g through the lower triangle only
 * (every stored element is used for a row sum and a column sum: 4 n^2 bytes per GEMV pass) and applies the
 * recorded updates to the lower triangle only (8 n^2 bytes per 8 updates; the upper half is mirrored back
 * before anything observes Q).  An equal-block row shard keeps the full-row passes, so its bits differ from an
 * unsharded handle's at depth 8 (both stay within the parity tolerance; shards agree with each other bit for
 * bit); a symmetric row shard (below) runs the lower-triangle schedule on its trapezoid.
 * depth = 16 / 24: the same with 16 / 24 recorded updates per apply pass ((4 + 8 / depth) n^2 bytes per update);
 * lower-triangle schedule only (ELLHIP_E_INVALID otherwise).  Depth 24 is the fastest schedule on MI355X where it
 * exists and what a new handle starts with there: its apply pass is ONE rank-24 update on the FP64 matrix cores
 * (k_apply_mfma, 0.43 ms at n = 16384 whatever the depth; the vector kernel k_apply_lower needs 0.40 ms for 16 updates
 * and 0.62 ms for 24).  On the matrix cores an element receives its updates as a chain of fused multiply-adds (one
 * rounding per update where the reference has two): inside the 1e-10 contract, not the reference's rounding sequence;
 * ELLHIP_OPT_APPLY_KERNEL = 1 selects the vector kernel and that sequence at any depth. */
int ellhip_set_defer_depth(ellhip_space *s, int depth);
int ellhip_defer_depth(const ellhip_space *s);
/* Apply whatever the deferred schedule has recorded so far (and a shrink a pipelined cut left pending) now,
 * asynchronously on the handle's stream.  Never needed for correctness (every observer of Q does it itself);
 * bench.py calls it at both ends of its timed region so that the region pays for ALL of its updates. */
int ellhip_flush(ellhip_space *s);
/* Queue index of the cut whose GEMV is in place (primed, not yet cut), or -1.  ellhip_flush and the observers of
 * Q (get_mq, clone, mode switches) keep a primed gradient valid on an unsharded handle (its Q_base*g is recomputed
 * when they apply recorded updates); on a row shard they DROP the prime instead, because the recomputation needs
 * the owner's collective -- a multi-GPU driver asks here after such a call and primes again if need be. */
int64_t ellhip_queue_primed(const ellhip_space *s);
/* Symmetric row shard (multi-GPU, deferred schedule only).  Call once after ellhip_create_shard, together with
 * ellhip_set_defer_depth(h, 8), on every rank; shard boundaries must be multiples of 64 and n even.  The GEMV
 * of each cut then reads only this shard's LOWER trapezoid (columns up to each row's diagonal) and leaves in the
 * gt buffer this shard's PARTIAL sums for all n entries (row sums for its rows, column sums for the columns left
 * of them, zeros to the right): the caller adds the shards' vectors with ONE all-reduce (sum) instead of the
 * all-gather of the row-block schedule.  Choosing the boundaries at n*sqrt(r/P) balances the trapezoids, so each
 * GPU moves 5 n^2 / P bytes per update instead of 9 n^2 / P.  The apply passes touch the lower trapezoid only
 * too: ellhip_get_mq on such a shard returns rows that are current up to their diagonal (the mirrored elements
 * live on other ranks).  Depth-1 schedules and no_defer_trick are refused (ELLHIP_E_STATE). */
int ellhip_set_shard_symmetric(ellhip_space *s, int flag);

/* ---- options ----------------------------------------------------------------------------------
 * Schedule / kernel-form choices a host may want to pin (A/B measurements, bit-identity tests between forms, a
 * device on which a form misbehaves).  Every option has the measured-best value as its default; nothing is read
 * from the environment.  ellhip_set_option changes ONE handle (Q is made current first, so it is legal at any
 * point between updates); ellhip_set_default_option changes what handles created LATER in this process start
 * with (process-wide, call it before creating handles from other threads).
 *   key                          values     default  meaning
 *   ELLHIP_OPT_AUTO_DEFER        0 / 1      1        default only: 0 = every new handle starts at depth 1 (the
 *                                                    reference's data flow) instead of 16 / 8 by size
 *   ELLHIP_OPT_SYMV              0 / 1      1        Ell: lower-triangle GEMV on the recorded schedule (4 n^2 bytes)
 *   ELLHIP_OPT_SYMV_MIN_N        >= 512     5120     Ell: smallest n of an unsharded handle that takes it (synchronous
 *                                                    updates break even there, queue runs gain 5-8x: tools/midsize_sweep.py)
 *   ELLHIP_OPT_APPLY_LOWER       0 / 1      1        Ell: apply passes touch the lower triangle only (8 n^2 bytes)
 *   ELLHIP_OPT_APPLY_KERNEL      -1 .. 2    -1       Ell, lower-triangle apply pass (-1 = 2 at depth 24, else 1): 2 = k_apply_mfma (the recorded updates as ONE rank-NP
 *                                                    update on the FP64 matrix cores: one rounding per update and element where the
 *                                                    reference has two -- inside the 1e-10 contract, not bit-identical to 0 / 1),
 *                                                    1 = k_apply_lower (16-row tiles, the reference's roundings), 0 = k_sweep_apply (depth 8)
 *   ELLHIP_OPT_FUSE_DOTS         0 / 1      1        Ell: the scalar stage's dot products come out of the GEMV's launch
 *   ELLHIP_OPT_RESIDENT          0 / 1 / 2  1        Ell: ellhip_queue_run / _run_fused of >= 4 cuts park the lower triangle in the
 *                                                    chip's register files and run the whole batch in ONE persistent launch
 *                                                    (n <= 4224 on a 256-CU device: 9 us instead of 39 us per update at
 *                                                    n = 4096); 0 = always the streamed schedules.  The launch is
 *                                                    cooperative (1: the grid is co-resident or the launch is refused; 2: a
 *                                                    plain launch, same kernel), the batches of one process are serialised
 *                                                    per device, and
 *                                                    one batch is synchronous: a refused launch or an in-launch wait that
 *                                                    gave up leaves Q, xc and the scalars at their pre-batch values, the
 *                                                    batch is rerun on the streamed schedule inside the same call and the
 *                                                    handle reads 0 here from then on
 *   ELLHIP_OPT_RESIDENT_FAULT    -1, >= 0   -1       Ell, per handle, TEST HOOK: one workgroup abandons every resident batch
 *                                                    at this cut of it as if its wait had timed out (exercises the above)
 *   ELLHIP_OPT_RESIDENT_ABANDONED  read only          Ell, per handle: resident batches abandoned and rerun so far
 *   ELLHIP_OPT_OVERLAP           0 / 1 / 2  1        Ell, ellhip_queue_run_fused on the lower-triangle schedule: the NEXT queued
 *                                                    cut's GEMV (LOOKAHEAD 1) or the next GROUP's products (LOOKAHEAD > 3) --
 *                                                    they read Q_base, which the cuts being taken do not change -- are issued
 *                                                    on a second stream: beside the current reduction + scalar stage
 *                                                    (LOOKAHEAD 1), behind the current group's reductions and beside the
 *                                                    rest of its stage (LOOKAHEAD > 3); same results as 0 to the bit (same
 *                                                    kernels, operands and summation order).  2: the same kernels in the
 *                                                    same order on ONE stream (what the tests compare 1 against)
 *   ELLHIP_OPT_LOOKAHEAD         1 .. 32    32       Ell, ellhip_queue_run_fused on the lower-triangle schedule: the GEMVs of up to
 *                                                    this many consecutive QUEUED cuts are formed in one pass over Q_base
 *                                                    (they all refer to the same matrix until the next apply pass):
 *                                                    (4 / L) n^2 bytes per update instead of 4 n^2.  L <= 3: vector-ALU
 *                                                    kernel, bit-identical to 1; L > 3 (n a multiple of 64): FP64 matrix
 *                                                    cores, y differs by a few ulp from the other schedules (own
 *                                                    association; inside the 1e-10 contract); groups of 17 .. 32 ride on
 *                                                    ONE pass with two 16-wide column tiles (the products are those of
 *                                                    two 16-wide passes to the bit).  Only a queue knows the
 *                                                    next gradients: ellhip_update and the prime / cut / commit calls are
 *                                                    unaffected.  The group runs need about n^2 doubles of extra device
 *                                                    memory -- as much as the matrix (2 x 32 sets of partial sums of
 *                                                    n^2 / 62 doubles each); a handle for which that does not
 *                                                    fit continues with 1 (and reports 1 here)
 *   ELLHIP_OPT_QUEUE_DEPTH       0 / 48     48       Ell, depth 24, LOOKAHEAD > 3: inside one ellhip_queue_run_fused call the
 *                                                    recorded updates may pile up to 48 before an apply pass (the group
 *                                                    stage is sized for it): half as many apply passes; on return fewer
 *                                                    than the handle's depth are left, as everywhere else.  0: never
 *                                                    more than the handle's depth
 *   ELLHIP_OPT_STABLE_SOLVE      0 .. 3     3        EllStable: 0 = one launch per 128-block (no in-launch waits),
 *                                                    1 = persistent solves, 2 = persistent + helper workgroups,
 *                                                    3 = 2 on the MIRRORED layout: the handle's private buffer holds the
 *                                                    factor on both sides of the diagonal, the scratch triangle
 *                                                    (src/ell_stable.rs:66) is not kept and the factor update (:107-121)
 *                                                    is applied by the next forward / backward solve to the tiles they
 *                                                    load -- 16 n^2 bytes per update instead of 20 n^2, no third pass;
 *                                                    ellhip_get_mq / ellhip_clone see the reference's buffer (rebuilt
 *                                                    exactly: identical bits to 0 / 1 / 2).  Where the helper form does
 *                                                    not fit the device (n > 16384 on 256 CUs) 3 runs as 1
 *   ELLHIP_OPT_STABLE_FACTOR     0 / 1 / 2  2        EllStable factor update: 0 = tile kernel reading the scratch
 *                                                    triangle (the reference's data flow, 12 n^2 bytes), 1 = row kernel
 *                                                    from U alone (8 n^2) beside the backward solve, 2 = pulled inside
 *                                                    the helped backward solve's launch
 *   ELLHIP_OPT_PAD               -1 .. 4096 -1       default only: extra doubles per row of Q (-1 = by size)
 *   ELLHIP_OPT_LP_GRID           0 .. 65536 0        default only: workgroups per LowpassOracle scan launch (0 = by size)
 *   ELLHIP_OPT_LP_WIDE           -1 / 0 / 1 -1       default only: column-split LowpassOracle scan kernel (-1 = by size)
 *   ELLHIP_OPT_BATCH_THREADS     0/64/128/256 0      default only: threads per workgroup of the batched engine
 * All forms of one option produce identical bits, except SYMV / SYMV_MIN_N (the lower-triangle GEMV sums Q*g in
 * another association than the full-row GEMV), FUSE_DOTS on the lower-triangle schedule (dot products summed per
 * 128 columns instead of per slice) and RESIDENT (Q*g summed per 4 x 2 register block and super-tile): there results
 * agree to ~1e-15, see "deferred shrink" above. */
#define ELLHIP_OPT_AUTO_DEFER 1
#define ELLHIP_OPT_SYMV 2
#define ELLHIP_OPT_SYMV_MIN_N 3
#define ELLHIP_OPT_APPLY_LOWER 4
#define ELLHIP_OPT_APPLY_KERNEL 5
#define ELLHIP_OPT_FUSE_DOTS 6
#define ELLHIP_OPT_STABLE_SOLVE 7
#define ELLHIP_OPT_STABLE_FACTOR 8
#define ELLHIP_OPT_PAD 9
#define ELLHIP_OPT_LP_GRID 10
#define ELLHIP_OPT_LP_WIDE 11
#define ELLHIP_OPT_BATCH_THREADS 12
#define ELLHIP_OPT_RESIDENT 13
#define ELLHIP_OPT_OVERLAP 14
#define ELLHIP_OPT_LOOKAHEAD 15
#define ELLHIP_OPT_QUEUE_DEPTH 16
#define ELLHIP_OPT_RESIDENT_FAULT 17
#define ELLHIP_OPT_RESIDENT_ABANDONED 18
#define ELLHIP_OPT_STABLE_MIRRORED 19   /* read only, EllStable: 1 while the handle's buffer is in the mirrored layout */
#define ELLHIP_OPT_STAGE_DIRECT 20      /* default only (0 / 1, default 1; per handle: read only): on a large-BAR system the host
                                           writes each gradient straight into (fine-grained) device memory instead of into a
                                           pinned buffer a kernel then pulls over PCIe */
int ellhip_set_option(ellhip_space *s, int key, int64_t value);
int ellhip_get_option(const ellhip_space *s, int key, int64_t *value);
int ellhip_set_default_option(int key, int64_t value);
int ellhip_default_option(int key, int64_t *value);

/* ---- device-resident cut queue (benchmarks, replay of recorded cut sequences) ---------------
 * Uploads k cuts once; run/begin/end then execute them without touching host memory, stopping
 * (all later cuts become no-ops with status ELLHIP_UNKNOWN) at the first non-Success one, like the
 * drivers do (src/cutting_plane.rs:222,308).  grads: k*n doubles. */
int ellhip_queue_upload(ellhip_space *s, int64_t k, const int32_t *kinds, const double *grads,
                        const double *beta0, const int32_t *has_beta1, const double *beta1);
/* Enqueue cuts [first, first+count) on the stream; asynchronous (except resident batches, ELLHIP_OPT_RESIDENT,
 * which have finished when the call returns).  ellhip_queue_run uses the
 * two-pass schedule (GEMV pass + rank-1 pass per cut); ellhip_queue_run_fused the pipelined one
 * (one pass per cut: the shrink of cut i fused with the GEMV of cut i+1). Same results.
 * On a handle that records its updates (lower-triangle schedule: unsharded, even n >= 5120 by default)
 * ellhip_queue_run_fused also uses that the queue holds the NEXT gradients: the products Q_base g of up to
 * ELLHIP_OPT_LOOKAHEAD consecutive queued cuts are formed in one pass over the matrix and their scalar stages run as
 * one group ("options" above; results to ~1e-15 of the cut-by-cut schedules, to the bit for LOOKAHEAD <= 3).  It may
 * use a second stream of its own beside the handle's; everything it issued there has been joined to the handle's
 * stream when it returns. */
int ellhip_queue_run(ellhip_space *s, int64_t first, int64_t count);
int ellhip_queue_run_fused(ellhip_space *s, int64_t first, int64_t count);
/* Phase-wise forms of one queued cut for the multi-GPU schedule; all asynchronous.  After every call
 * that ran a GEMV (begin, prime, commit with next_index >= 0) the caller all-gathers ellhip_gt_dev.
 *   two-pass:   queue_begin(i) -> gather -> queue_end(i)
 *   pipelined:  queue_prime(first) -> gather -> { queue_cut(i) -> queue_commit(i, i+1 or -1) -> gather } */
int ellhip_queue_begin(ellhip_space *s, int64_t index);
int ellhip_queue_end(ellhip_space *s, int64_t index);
int ellhip_queue_prime(ellhip_space *s, int64_t index);
int ellhip_queue_cut(ellhip_space *s, int64_t index);
int ellhip_queue_commit(ellhip_space *s, int64_t index, int64_t next_index);
/* Waits for the stream, then copies per-cut status (int32) and tsq (double) for all k cuts. */
int ellhip_queue_results(ellhip_space *s, int32_t *status_out, double *tsq_out);

/* ---- streams, sync, timing ------------------------------------------------------------------ */
/* hipStream_t to issue on (NULL = the handle's own stream). */
int ellhip_set_stream(ellhip_space *s, void *hip_stream);
int ellhip_synchronize(ellhip_space *s);
/* Per-kernel HIP-event timing: when enabled every kernel launch is bracketed by events on the
 * launch stream.  ellhip_profile_read waits for them and returns accumulated milliseconds and
 * launch counts per kernel class, then resets.  Classes: 0 = GEMV pass (Q*g), 1 = scalar stage,
 * 2 = rank-1 pass, 3 = EllStable forward, 4 = EllStable backward, 5 = EllStable factor update,
 * 6 = fused pass (rank-1 of cut k + GEMV of cut k+1), 7 = deferred apply pass (8 recorded updates),
 * 8 = deferred apply pass fused with the next GEMV, 9 = symmetric GEMV pass (lower triangle only,
 * 4 n^2 bytes; deferred mode on an unsharded handle), 10 = the partial-sum reduction that follows it,
 * 11 = LowpassOracle scan (ellhip_lowpass.h), 12 = LowpassOracle finish (cut assembly), 13 = resident queue run (one
 * launch per batch of cuts, the matrix parked on-chip). */
#define ELLHIP_NKERNEL_CLASSES 14
int ellhip_profile_enable(ellhip_space *s, int flag);
int ellhip_profile_read(ellhip_space *s, double *ms_out, int64_t *count_out);

/* ---- misc ----------------------------------------------------------------------------------- */
/* Number of HIP devices visible (0 if none / no driver). */
int ellhip_device_count(void);
/* Text of the last failure on this thread (never NULL). */
const char *ellhip_last_error(void);
/* "ellhip <version> gfx950" */
const char *ellhip_version(void);

#ifdef __cplusplus
}
#endif
#endif
