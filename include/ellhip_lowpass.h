/*
 * ellhip_lowpass.h -- C ABI of the device-side LowpassOracle and of the device-resident
 * cutting-plane loops built on it (libellhip.so).
 *
 * Reference: `LowpassOracle` (src/oracles/lowpass_oracle.rs:7-151), the reference's only large-n,
 * parallel-cut oracle: a 15n x n table `spectrum` and, per call, a round-robin walk over three
 * frequency bands that returns a cut at the first violated constraint.  A Rust binding keeps the
 * struct's public fields on the device behind this handle and implements
 * `OracleFeas<Arr>` / `OracleOptim<Arr>` (src/cutting_plane.rs:119-136) by calling
 * ellhip_lowpass_assess_feas / ellhip_lowpass_assess_optim.
 *
 * Same conventions as ellhip.h: host buffers owned by the caller, plain pointers and sizes, 0 = ok,
 * negative = ELLHIP_E_*, no CPU fallback.
 */
#ifndef ELLHIP_LOWPASS_H
#define ELLHIP_LOWPASS_H

#include "ellhip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ellhip_lowpass ellhip_lowpass;

/* LowpassOracle::new(ndim, wpass, wstop, lp_sq, up_sq, sp_sq) (src/oracles/lowpass_oracle.rs:23-53).
 * `spectrum`: the caller's table, row-major (15*ndim) x ndim, or NULL to have it computed exactly as
 * the reference does (linspace(0, pi, 15 ndim), row i = [1, 2cos(w_i), 2cos(2 w_i), ...], host libm).
 * The table lives in HBM (15 n^2 * 8 bytes: 2 GiB at n = 4096).  create_lowpass_case(ndim)
 * (:153-167) is this call with its constants; note that they give lp_sq > up_sq, so
 * the first cut has beta1 < beta0 and every search space answers NoSoln at iteration 0. */
int ellhip_lowpass_create(ellhip_lowpass **out, int64_t ndim, double wpass, double wstop, double lp_sq,
                          double up_sq, double sp_sq, const double *spectrum, int device);
void ellhip_lowpass_destroy(ellhip_lowpass *o);

/* OracleFeas::assess_feas(&mut self, x) -> Option<(Arr, ParallelCut)> (:58-133).
 * Returns 1 and fills grad_out[ndim], beta0, has_beta1, beta1 = Some((grad, ParallelCut(beta0,
 * has_beta1 ? Some(beta1) : None))); returns 0 = None (x is feasible); negative = failure.
 * The walk evaluates the rows the reference evaluates (plus at most one grid-wide round of 16-row
 * chunks) and returns the cut of the FIRST violated constraint in the reference's visiting order. */
int ellhip_lowpass_assess_feas(ellhip_lowpass *o, const double *x, double *grad_out, double *beta0,
                               int *has_beta1, double *beta1);

/* OracleOptim::assess_optim(&mut self, x, &mut sp_sq) -> ((Arr, ParallelCut), bool) (:139-150).
 * gamma_inout is `sp_sq`; *shrunk is the bool.  Returns 1 (a cut is always produced), or
 * ELLHIP_E_STATE when x is feasible but no stopband row exists (the reference would panic). */
int ellhip_lowpass_assess_optim(ellhip_lowpass *o, const double *x, double *gamma_inout, double *grad_out,
                                double *beta0, int *has_beta1, double *beta1, int *shrunk);

/* The struct's public fields: ints7 = {more_alt, idx1, idx2, idx3, kmax, nwpass, nwstop},
 * doubles2 = {fmax, sp_sq}.  After a call that returned a cut from the passband, fmax / kmax keep
 * their previous values; otherwise they cover the stopband rows the walk visited (:86-103). */
int ellhip_lowpass_state(ellhip_lowpass *o, int32_t *ints7, double *doubles2);
/* Work measure: number of rows the reference's walk visits (row.x products it computes), summed over
 * all calls since creation / the last reset; negative = failure.  The device reads these rows plus at
 * most one grid-wide round of 16-row chunks per call. */
int64_t ellhip_lowpass_rows_visited(ellhip_lowpass *o, int reset);
/* The table, row-major (15*ndim) x ndim (the `spectrum` field). */
int ellhip_lowpass_get_spectrum(ellhip_lowpass *o, double *out);

/* cutting_plane_optim(&mut omega, &mut space, &mut gamma, &Options{max_iters, tolerance})
 * (src/cutting_plane.rs:286-313) with omega = this oracle and space = an UNSHARDED ellhip_space
 * (Ell at any defer depth, or EllStable), run entirely on the device: the centre never leaves HBM,
 * the oracle writes the gradient and the cut values where the update reads them, x_best / gamma /
 * the iteration count / the stop test (status != Success || tsq < tolerance) live on the device, and
 * the host enqueues iterations in batches of 64 and looks at the loop state once per batch.  The
 * oracle for iteration k+1 runs between the scalar stage and the shrink of iteration k, so at defer
 * depth 1 the shrink carries the next GEMV (16 n^2 bytes per update instead of 24 n^2).
 * Outputs: x_best_out[ndim] (written when *has_best_out), *niter_out, *gamma_inout: exactly the
 * reference's (x_best, niter) and gamma.  The space and the oracle are left in the state the
 * reference loop leaves them in (the update that hits the tolerance is complete). */
int ellhip_lowpass_optim(ellhip_space *s, ellhip_lowpass *o, double *gamma_inout, int64_t max_iters, double tol,
                         double *x_best_out, int *has_best_out, int64_t *niter_out);
/* cutting_plane_feas (src/cutting_plane.rs:205-227), same way: *feasible_out and x_out = Some(x). */
int ellhip_lowpass_feas(ellhip_space *s, ellhip_lowpass *o, int64_t max_iters, double tol, double *x_out,
                        int *feasible_out, int64_t *niter_out);

#ifdef __cplusplus
}
#endif
#endif
