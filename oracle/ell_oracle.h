/*
 * ell_oracle.h -- CPU ORACLE for the ellipsoid-update hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's arithmetic (luk036/ellalgo-rs 0.1.7) for
 * Ell::update_core, EllStable::update_core and the EllCalc coefficient formulas, keeping the
 * reference's operation and loop order so that results are bit-comparable with the Rust code.
 * It is the CHECKER the HIP path is compared against.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; nothing under ellalgo-rs_amd/ (the product) does.
 *
 * Pinning: tests/test_oracle_pins.py checks this file against every known answer the reference's
 * own tests hold for the path (src/ell.rs:247-354, src/ell_calc.rs:942-1186,
 * src/ell_calc_additional_tests.rs, src/ell_stable.rs:217-307 and the pinned iteration counts
 * listed in SURVEY.md section 8c), and tests/golden/ holds vectors produced together with the
 * reference's python_ai Ell._update_core dense arithmetic.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: no FMA contraction, as in rustc).
 */
#ifndef ELL_ORACLE_H
#define ELL_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* CutStatus in the reference's declaration order (src/cutting_plane.rs:31-37). */
enum { ORC_SUCCESS = 0, ORC_NOSOLN = 1, ORC_NOEFFECT = 2, ORC_UNKNOWN = 3 };
/* Which SearchSpace entry point was called (src/cutting_plane.rs:161-179). */
enum { ORC_CUT_BIAS = 0, ORC_CUT_CENTRAL = 1, ORC_CUT_Q = 2 };

/* EllCalcCore + EllCalc (src/ell_calc.rs:22-29, 627-631). */
typedef struct {
    double n_f, n_plus_1, half_n, inv_n, cst1, cst2;
    int use_parallel_cut;
} orc_calc;

void orc_calc_init(orc_calc *c, int64_t n);

/* EllCalcCore formulas; out = {rho, sigma, delta}. */
void orc_core_parallel_bias_cut_fast(const orc_calc *c, double b0, double b1, double tsq,
                                     double b0b1, double eta, double out[3]);
void orc_core_parallel_bias_cut(const orc_calc *c, double b0, double b1, double tsq, double out[3]);
void orc_core_parallel_central_cut(const orc_calc *c, double b1, double tsq, double out[3]);
void orc_core_bias_cut_fast(const orc_calc *c, double beta, double tau, double eta, double out[3]);
void orc_core_bias_cut(const orc_calc *c, double beta, double tau, double out[3]);
void orc_core_central_cut(const orc_calc *c, double tsq, double out[3]);

/* EllCalc gatekeepers; return CutStatus, out = {rho, sigma, delta}. */
int orc_calc_bias_cut(const orc_calc *c, double beta, double tsq, double out[3]);
int orc_calc_bias_cut_q(const orc_calc *c, double beta, double tsq, double out[3]);
int orc_calc_central_cut(const orc_calc *c, double tsq, double out[3]);
int orc_calc_parallel_bias_cut(const orc_calc *c, double b0, double b1, double tsq, double out[3]);
int orc_calc_parallel_q(const orc_calc *c, double b0, double b1, double tsq, double out[3]);
int orc_calc_parallel_central_cut(const orc_calc *c, double b1, double tsq, double out[3]);
/* CutType dispatch of src/ell.rs:182-210: has_b1 == 0 covers SingleCut and ParallelCut(b0, None). */
int orc_calc_dispatch(const orc_calc *c, int kind, double b0, int has_b1, double b1, double tsq,
                      double out[3]);

/* Ell (src/ell.rs:9-16). mq is n*n row-major. */
typedef struct {
    int64_t n;
    double *mq;
    double *xc;
    double kappa, tsq;
    int no_defer_trick;
    orc_calc helper;
    double *gt; /* scratch, n */
} orc_ell;

/* mq == NULL && diag == NULL -> identity; diag != NULL -> diag(diag); else copy of mq. */
orc_ell *orc_ell_new(int64_t n, double kappa, const double *mq, const double *diag, const double *xc);
orc_ell *orc_ell_clone(const orc_ell *e);
void orc_ell_free(orc_ell *e);
int orc_ell_update(orc_ell *e, int kind, const double *grad, double b0, int has_b1, double b1);
/* Same arithmetic as orc_ell_update but the symmetric rank-1 is done row-wise over the full
 * matrix with the (ratio*gt[max])*gt[min] form (bit-identical for symmetric Q, no strided
 * mirror stores). Used to cross-check the identity the HIP kernel relies on. */
int orc_ell_update_rowwise(orc_ell *e, int kind, const double *grad, double b0, int has_b1, double b1);
/* NOT the reference's loop: orc_ell_update_rowwise with its rows spread over OpenMP threads (bit-identical to
 * it); for the "all cores" line of bench.py's cpu_baseline only. */
int orc_ell_update_rowwise_mt(orc_ell *e, int kind, const double *grad, double b0, int has_b1, double b1);
/* Size of the OpenMP team of the _mt variant (0 = leave the runtime's default); returns the size in force. */
int orc_set_num_threads(int nthreads);
double orc_ell_kappa(const orc_ell *e);
double orc_ell_tsq(const orc_ell *e);
double *orc_ell_mq(orc_ell *e);
double *orc_ell_xc(orc_ell *e);
void orc_ell_set_no_defer_trick(orc_ell *e, int flag);
void orc_ell_set_use_parallel_cut(orc_ell *e, int flag);

/* Row-block pieces of the same update, for the row-partitioned multi-GPU schedule (tests only):
 * phase 1 computes gt[row0 .. row0+nrows) from the local rows, phase 2 applies the scalar stage
 * (on the full gt) and the rank-1 to the local rows. mq_local is nrows*n row-major. */
void orc_rows_gemv(int64_t n, int64_t row0, int64_t nrows, const double *mq_local,
                   const double *grad, double *gt_full);

/* B independent Ell spaces (identity, kappa0, xc = 0), K cuts each: a loop over orc_ell_update.
 * kinds/b0/has_b1/b1 [K][B], grads [K][B][n]; outputs may be NULL.  Returns the number of Success cuts. */
int64_t orc_ell_batch_run(int64_t B, int64_t n, int64_t K, const int32_t *kinds, const double *grads,
                          const double *b0, const int32_t *has_b1, const double *b1, double kappa0,
                          int32_t *status_out, double *mq_out, double *xc_out, double *kappa_out);

/* EllStable (src/ell_stable.rs:9-15): one n*n buffer, diag = D entries, strict upper = L^T,
 * strict lower = scratch. corrected != 0 selects the mathematically consistent variant
 * (back-substitution on the factor, running v in the rank-one update); 0 = as the reference. */
typedef struct {
    int64_t n;
    double *mq;
    double *xc;
    double kappa, tsq;
    int corrected;
    orc_calc helper;
    double *w, *z, *gg, *q; /* scratch, n each */
} orc_ellstable;

orc_ellstable *orc_ellstable_new(int64_t n, double kappa, const double *mq, const double *diag,
                                 const double *xc);
orc_ellstable *orc_ellstable_clone(const orc_ellstable *e);
void orc_ellstable_free(orc_ellstable *e);
int orc_ellstable_update(orc_ellstable *e, int kind, const double *grad, double b0, int has_b1,
                         double b1);
double orc_ellstable_kappa(const orc_ellstable *e);
double orc_ellstable_tsq(const orc_ellstable *e);
double *orc_ellstable_mq(orc_ellstable *e);
double *orc_ellstable_xc(orc_ellstable *e);
void orc_ellstable_set_corrected(orc_ellstable *e, int flag);

#ifdef __cplusplus
}
#endif
#endif
