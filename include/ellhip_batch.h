/*
 * ellhip_batch.h -- C ABI of the batched small-n engine (libellhip.so).
 *
 * B independent `Ell` search spaces (src/ell.rs:9-16) of one dimension n <= 128, side by side in HBM, updated
 * together: one call applies K cuts to each of them, one workgroup per ellipsoid with its matrix in LDS.
 * It covers the reference's small-n usage, where a single ellipsoid can never fill a GPU:
 * `BSearchAdaptor::assess_bs` clones the space for every probe (src/cutting_plane.rs:403-419),
 * the randomized sweeps and BASELINE config 1 (n = 16) run many tiny problems.  A binding keeps a
 * `Vec<Ell>` behind this handle and calls ellhip_batch_update where it would loop over
 * `space[b].update_bias_cut(&cut[b])`.
 *
 * Every step follows the reference's statement order (row-wise left folds, rank-1 over j <= i with the mirror
 * store), so the results are bit-identical to the CPU arithmetic, not merely within 1e-10.
 * Same conventions as ellhip.h: host buffers owned by the caller unless the name ends in _dev, 0 = ok,
 * negative = ELLHIP_E_*, no CPU fallback.
 */
#ifndef ELLHIP_BATCH_H
#define ELLHIP_BATCH_H

#include "ellhip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ellhip_batch ellhip_batch;

#define ELLHIP_BATCH_NMAX 128

/* B ellipsoids of dimension n.  kappa: B values, or NULL = 1.0 each.  mq: B*n*n row-major (Ell::new_with_matrix,
 * src/ell.rs:31-41; need not be symmetric), or NULL with diag: B*n (Ell::new, :55-57), or both NULL = identity
 * (Ell::new_with_scalar with kappa = val, :71-73).  xc: B*n, or NULL = 0. */
int ellhip_batch_create(ellhip_batch **out, int64_t B, int64_t n, const double *kappa, const double *mq,
                        const double *diag, const double *xc, int device);
/* B clones of one unsharded Ell handle (`self.space.clone()` per probe, src/cutting_plane.rs:410). */
int ellhip_batch_from_space(ellhip_batch **out, const ellhip_space *space, int64_t B);
void ellhip_batch_destroy(ellhip_batch *h);

/* K cuts for each ellipsoid, applied in order k = 0..K-1, exactly as K successive calls of
 * SearchSpace::update_{bias,central}_cut / update_q (kind as in ellhip_update) on each of the B spaces:
 * kinds, beta0, has_beta1, beta1 are [K][B]; grads is [K][B][n]; status_out [K][B] receives the CutStatus of
 * every call (a failed cut leaves its ellipsoid untouched except tsq, src/ell.rs:105-109, and the following cuts
 * of that ellipsoid are still applied, as they would be by a caller that keeps calling); tsq_out [K][B] or
 * NULL.  Synchronous at return. */
int ellhip_batch_update(ellhip_batch *h, int64_t K, const int32_t *kinds, const double *grads, const double *beta0,
                        const int32_t *has_beta1, const double *beta1, int32_t *status_out, double *tsq_out);
/* Same with every array already in HBM (device pointers), asynchronous on the handle's stream: the form a
 * device-side producer of cuts uses, and the one bench.py times. */
int ellhip_batch_update_dev(ellhip_batch *h, int64_t K, const int32_t *kinds_dev, const double *grads_dev,
                            const double *beta0_dev, const int32_t *has_beta1_dev, const double *beta1_dev,
                            int32_t *status_out_dev, double *tsq_out_dev);
int ellhip_batch_synchronize(ellhip_batch *h);
void *ellhip_batch_stream(ellhip_batch *h); /* hipStream_t the handle issues on */

/* state of all ellipsoids: xc [B][n], mq [B][n][n], kappa [B], tsq [B] */
int ellhip_batch_get_xc(ellhip_batch *h, double *out);
int ellhip_batch_set_xc(ellhip_batch *h, const double *xc);
int ellhip_batch_get_mq(ellhip_batch *h, double *out);
int ellhip_batch_get_kappa(ellhip_batch *h, double *out);
int ellhip_batch_get_tsq(ellhip_batch *h, double *out);
int64_t ellhip_batch_size(const ellhip_batch *h);
int64_t ellhip_batch_ndim(const ellhip_batch *h);
int ellhip_batch_set_no_defer_trick(ellhip_batch *h, int flag);
int ellhip_batch_set_use_parallel_cut(ellhip_batch *h, int flag);

#ifdef __cplusplus
}
#endif
#endif
