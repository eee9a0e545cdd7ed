/*
 * ellhip_sharded.h -- the multi-GPU form of the Ell search space behind the SAME boundary (SURVEY.md 8b
 * `create(..., ngpus)`, 8e): Q is partitioned by row blocks over the GPUs of one node, ONE process per GPU, and every
 * update assembles the full Q*g with ONE collective on an n-vector over RCCL (xGMI) issued by the library itself on
 * the handle's stream -- a Rust / C / C++ host needs no collective library of its own and no Python:
 *
 *     phase 1  local pass               gt[R_r] = Q[R_r, :] * g                       (ellhip_update_begin)
 *     exchange ncclAllGather of gt, in place (equal row blocks, n / P doubles per rank), or
 *              ncclAllReduce(sum) of the n partial sums (symmetric row shards)      (the ONLY collective)
 *     phase 2  scalar stage, redundantly on every rank (identical bits: fixed reduction shapes), then the shrink of
 *              the local rows                                                         (ellhip_update_end)
 *
 * What it replaces in the reference: nothing -- ellalgo-rs is single-threaded; `impl SearchSpace for
 * Ell` (src/ell.rs:140-180) keeps its signatures, the trait object just lives on P ranks that make the same calls with
 * the same arguments (the oracle is evaluated redundantly or its cut is broadcast by the host; the space never calls
 * the oracle, src/cutting_plane.rs:129-136).
 *
 * RCCL is opened at run time (dlopen "librccl.so.1"; a copy already mapped by the process -- e.g. PyTorch-ROCm's -- is
 * reused), so libellhip.so itself loads where RCCL is absent; the calls below then fail with ELLHIP_E_NORCCL.
 * Conventions as in ellhip.h: host buffers in / out, 0..3 = CutStatus from the update calls, negative = failure.
 * Every rank of the group must make the same sequence of calls (they are collective).
 */
#ifndef ELLHIP_SHARDED_H
#define ELLHIP_SHARDED_H

#include "ellhip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ELLHIP_E_NORCCL (-6)       /* librccl could not be opened, or one of its calls failed (ellhip_last_error) */
#define ELLHIP_NCCL_ID_BYTES 128   /* sizeof(ncclUniqueId) */

#define ELLHIP_SHARD_EQUAL_BLOCKS 0   /* rows [r n/P, (r+1) n/P); all-gather; every schedule (depth 1 / 8); n % P == 0 */
#define ELLHIP_SHARD_SYMMETRIC 1      /* boundaries at n sqrt(r/P) rounded to 64 rows: equal lower-trapezoid areas;
                                         lower-triangle GEMVs + apply passes, partial sums added by one all-reduce;
                                         recorded schedule (depth 8 / 16) only; n % 64 == 0, n / 64 >= P */

typedef struct ellhip_sharded ellhip_sharded; /* opaque: this rank's row block + the communicator */

/* Rows of rank `rank` under a partition (what ellhip_sharded_create uses; callers slice a matrix with it). */
int ellhip_sharded_partition(int64_t n, int nranks, int rank, int partition, int64_t *row0_out, int64_t *nrows_out);

/* Bootstrap for hosts that have no RCCL binding: rank 0 calls this and ships the 128 bytes to the other ranks by any
 * means it has (a file, a socket, MPI, ...); every rank then passes them to ellhip_sharded_create. */
int ellhip_sharded_unique_id(void *id_out /* ELLHIP_NCCL_ID_BYTES */);

/* This rank's part of an n-dimensional Ell spread over `nranks` processes.  mq_rows: THIS rank's row block
 * (nrows x n, row-major) of a symmetric matrix, or NULL for the identity / diag(diag) (Ell::new_with_scalar / new /
 * new_with_matrix, src/ell.rs:31-78).  Exactly one of `nccl_id` (the bytes from ellhip_sharded_unique_id: the library
 * creates -- and owns -- the communicator with ncclCommInitRank) and `nccl_comm` (an ncclComm_t the host already has;
 * it stays the host's) is given; with nranks == 1 both may be NULL (no collective is issued).
 * defer_depth: 1, 8 or 16 (ellhip_set_defer_depth); the symmetric partition needs 8 or 16. */
int ellhip_sharded_create(ellhip_sharded **out, int64_t n, double kappa, const double *mq_rows, const double *diag,
                          const double *xc, int device, int rank, int nranks, const void *nccl_id, void *nccl_comm,
                          int partition, int defer_depth);
void ellhip_sharded_destroy(ellhip_sharded *s);

/* The same with a HOST-SUPPLIED collective instead of RCCL (an MPI host, another transport, a test double that lets
 * several ranks share one GPU -- RCCL refuses two ranks per device).  The library calls the callbacks where it would
 * call ncclAllGather / ncclAllReduce: once per update, on the calling host thread, between the local pass and the
 * scalar stage.  `vec_dev` is this rank's n-vector in DEVICE memory; the local pass that produced it has been
 * ENQUEUED on `hip_stream` (a hipStream_t) and the work that consumes it will be enqueued there after the callback
 * returns: a callback either enqueues its own work on that stream, or synchronises it, exchanges, and returns.
 *   allgather  in place: this rank contributed vec_dev[offset, offset + count) (its rows; every rank has the same
 *              count, rank r's offset is r * count); on return the vector holds every rank's part
 *   allreduce  in place: on return vec_dev[0, count) holds the element-wise SUM over the ranks, the same bits on
 *              every rank (the scalar stage runs redundantly and must see identical input)
 * Both return 0 on success; anything else fails the update with ELLHIP_E_NORCCL.  Both callbacks are required. */
typedef int (*ellhip_allgather_fn)(void *ctx, double *vec_dev, int64_t offset, int64_t count, void *hip_stream);
typedef int (*ellhip_allreduce_fn)(void *ctx, double *vec_dev, int64_t count, void *hip_stream);
int ellhip_sharded_create_custom(ellhip_sharded **out, int64_t n, double kappa, const double *mq_rows, const double *diag,
                                 const double *xc, int device, int rank, int nranks, int partition, int defer_depth,
                                 ellhip_allgather_fn allgather, ellhip_allreduce_fn allreduce, void *ctx);
/* Replace the collective of an existing handle (both NULL: back to the handle's RCCL communicator). */
int ellhip_sharded_set_collective(ellhip_sharded *s, ellhip_allgather_fn allgather, ellhip_allreduce_fn allreduce, void *ctx);

/* SearchSpace (src/cutting_plane.rs:154-182); collective: every rank passes the same cut */
int ellhip_sharded_update(ellhip_sharded *s, int kind, const double *grad, double beta0, int has_beta1, double beta1);
double ellhip_sharded_tsq(const ellhip_sharded *s);
double ellhip_sharded_kappa(const ellhip_sharded *s);
int ellhip_sharded_get_xc(const ellhip_sharded *s, double *xc_out);   /* the full centre (replicated) */
int ellhip_sharded_set_xc(ellhip_sharded *s, const double *xc);
/* this rank's rows of Q (nrows x n).  Symmetric partition: a row is current up to its diagonal only (the mirrored
 * elements live on other ranks). */
int ellhip_sharded_get_mq_rows(ellhip_sharded *s, double *rows_out);
int ellhip_sharded_set_defer_depth(ellhip_sharded *s, int depth);
int ellhip_sharded_flush(ellhip_sharded *s);

/* device-resident cut queue (benchmarks, replay): every rank uploads the same cuts */
int ellhip_sharded_queue_upload(ellhip_sharded *s, int64_t k, const int32_t *kinds, const double *grads,
                                const double *beta0, const int32_t *has_beta1, const double *beta1);
int ellhip_sharded_queue_run(ellhip_sharded *s, int64_t first, int64_t count);        /* two passes per cut */
/* pipelined: one pass over the local rows per cut and one collective per cut -- or, for symmetric shards with n a multiple of
 * 64 (ELLHIP_OPT_LOOKAHEAD > 3, the default), per GROUP of up to 32 queued cuts: their products Q_base g are formed in one
 * pass over the local trapezoid on the matrix cores, the group's partial vectors are added by ONE all-reduce (cuts x n
 * doubles through the same ellhip_allreduce_fn / ncclAllReduce), and the group's scalar stage runs on every rank
 * (include/ellhip.h "options", DESIGN.md sections 3.6 and 7).  Results agree with the cut-by-cut schedule to ~1e-15. */
int ellhip_sharded_queue_run_fused(ellhip_sharded *s, int64_t first, int64_t count);
int ellhip_sharded_queue_results(ellhip_sharded *s, int32_t *status_out, double *tsq_out);
int ellhip_sharded_synchronize(ellhip_sharded *s);

/* this rank's row-block handle (ellhip.h calls that are local: profiling, stream, ...); owned by `s` */
ellhip_space *ellhip_sharded_local(ellhip_sharded *s);

/* ONE process driving all P row blocks itself (several GPUs of a node from one host thread, or several blocks on one
 * GPU): the shard handles come from ellhip_create_shard with the rows of ellhip_sharded_partition(.., EQUAL_BLOCKS ..),
 * an update is ellhip_update_begin on every shard, then this call, then ellhip_update_end on every shard.  It is the
 * all-gather of the schedule above done with device-to-device copies: every shard's own rows of Q*g are copied into
 * the same rows of every other shard's vector (peer copies between devices).  Equal row blocks only. */
int ellhip_shards_exchange(ellhip_space *const *shards, int nshards);

#ifdef __cplusplus
}
#endif
#endif /* ELLHIP_SHARDED_H */
