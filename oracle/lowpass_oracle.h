/*
 * lowpass_oracle.h -- CPU ORACLE for the LowpassOracle row.  TEST INFRASTRUCTURE ONLY
 * (same rules as ell_oracle.h: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may
 * load it).
 *
 * Plain-C restatement of src/oracles/lowpass_oracle.rs:7-167 (struct, new, assess_feas, assess_optim,
 * create_lowpass_case) and of the two driver loops it is used with (src/cutting_plane.rs:205-227,
 * 286-313) over the oracle search spaces of ell_oracle.h, keeping the reference's statement order.
 *
 * Pinning: the reference's own tests for this oracle assert very little (src/oracles/
 * lowpass_oracle.rs:176-240, tests/stress_tests.rs:7-24: `is_some()`, finiteness, no counts) because
 * create_lowpass_case's constants give lp_sq > up_sq, so every run ends with NoSoln at iteration 0
 * (SURVEY F7).  tests/test_oracle_pins.py checks exactly that behaviour (cut at x = 0: row 0 negated,
 * beta = (lp_sq, up_sq), NoSoln, niter = 0, no x_best).  For the `corrected` constants (the ones the
 * reference's upstream Python package uses: passband ripple 1 +/- 0.025, stopband 0.125) NO reference
 * answer exists: those runs are "parity unpinned" -- the HIP path is compared with this restatement
 * only.
 */
#ifndef LOWPASS_ORACLE_H
#define LOWPASS_ORACLE_H

#include "ell_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int more_alt;
    int idx1;
    double *spectrum; /* mdim x ndim row-major (Vec<Arr>) */
    int64_t ndim, mdim;
    int nwpass, nwstop;
    double lp_sq, up_sq, sp_sq;
    int idx2, idx3;
    double fmax;
    int kmax;
    int64_t rows_visited; /* rows whose dot product was taken in the last call (work measure) */
} orc_lowpass;

orc_lowpass *orc_lowpass_new(int64_t ndim, double wpass, double wstop, double lp_sq, double up_sq,
                             double sp_sq);
void orc_lowpass_free(orc_lowpass *o);
/* out5 = {wpass, wstop, lp_sq, up_sq, sp_sq}.  corrected == 0: create_lowpass_case as written
 * (:153-167); 1: delta1 = 20 log10(1 + 0.025), delta2 = 20 log10(0.125). */
void orc_lowpass_case(int corrected, double out5[5]);
/* 1 = Some(cut) (g, b0, has_b1, b1 filled), 0 = None */
int orc_lowpass_assess_feas(orc_lowpass *o, const double *x, double *g, double *b0, int *has_b1,
                            double *b1);
/* always produces a cut; *shrunk = the bool; *gamma = sp_sq in/out.  Returns 1, or -1 if the
 * reference would index spectrum[-1] (feasible x, kmax == -1). */
int orc_lowpass_assess_optim(orc_lowpass *o, const double *x, double *gamma, double *g, double *b0,
                             int *has_b1, double *b1, int *shrunk);

/* cutting_plane_optim / cutting_plane_feas with this oracle; space_kind 0 = orc_ell, 1 =
 * orc_ellstable (`space` is the matching pointer).  Returns niter; *has_best / x_best as the
 * reference's Option<Arr>. */
int64_t orc_lowpass_cutting_plane_optim(orc_lowpass *o, int space_kind, void *space, double *gamma,
                                        int64_t max_iters, double tol, double *x_best, int *has_best,
                                        int *last_status);
int64_t orc_lowpass_cutting_plane_feas(orc_lowpass *o, int space_kind, void *space, int64_t max_iters,
                                       double tol, double *x_out, int *feasible, int *last_status);

#ifdef __cplusplus
}
#endif
#endif
