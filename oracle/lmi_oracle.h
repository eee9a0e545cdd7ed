/*
 * lmi_oracle.h -- CPU ORACLE for the LDLTMgr / LMI oracles row.  TEST INFRASTRUCTURE ONLY
 * (same rules as ell_oracle.h).
 *
 * Plain-C restatement, statement by statement, of src/oracles/ldlt_mgr.rs:3-141 (LDLTMgr: factor,
 * factor_with_allow_semidefinite, is_spd, witness, sym_quad, sqrt), src/oracles/lmi_oracle.rs:5-45 (LMIOracle)
 * and src/oracles/lmi0_oracle.rs:4-35 (LMI0Oracle).
 *
 * Pinning: tests/test_oracle_pins.py / tests/test_lmi_oracle_cpu.py check it against the reference's own known
 * answers: src/oracles/ldlt_mgr.rs:143-268 (chol1..chol9: SPD flags, pos tuples, witness values, sqrt),
 * tests/lmi_tests.rs:58-113 (cuts at given points) and :199-225 (x_best is Some, < 300 / < 400 iterations).
 */
#ifndef LMI_ORACLE_H
#define LMI_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int64_t pos0, pos1; /* pub pos: (usize, usize) */
    double *wit;        /* pub wit */
    int64_t ndim;
    double *storage;    /* ndim * ndim */
} orc_ldlt;

orc_ldlt *orc_ldlt_new(int64_t ndim);
void orc_ldlt_free(orc_ldlt *m);
/* get_elem closure as a callback */
typedef double (*orc_elem_fn)(void *ctx, int64_t i, int64_t j);
int orc_ldlt_factor(orc_ldlt *m, orc_elem_fn get, void *ctx);
int orc_ldlt_factor_semidefinite(orc_ldlt *m, orc_elem_fn get, void *ctx);
int orc_ldlt_factorize(orc_ldlt *m, const double *mat); /* mat: ndim*ndim row-major */
int orc_ldlt_is_spd(const orc_ldlt *m);
double orc_ldlt_witness(orc_ldlt *m);
double orc_ldlt_sym_quad(const orc_ldlt *m, const double *mat);
void orc_ldlt_sqrt(const orc_ldlt *m, double *r_out); /* ndim*ndim row-major upper triangular */

/* LMIOracle (mode 0: F(x) = B - sum x_k F_k, g_k = +v'F_k v) and LMI0Oracle (mode 1: F(x) = sum x_k F_k,
 * g_k = -v'F_k v).  mat_f: n matrices of m*m row-major, contiguous; mat_b: m*m or NULL (mode 1). */
typedef struct {
    int mode;
    int64_t n, m;
    double *mat_f, *mat_b;
    orc_ldlt *ldlt;
} orc_lmi;

orc_lmi *orc_lmi_new(int mode, int64_t n, int64_t m, const double *mat_f, const double *mat_b);
void orc_lmi_free(orc_lmi *o);
/* 1 = Some((g, SingleCut(ep))) with g[n], *ep filled; 0 = None */
int orc_lmi_assess_feas(orc_lmi *o, const double *x, double *g, double *ep);

#ifdef __cplusplus
}
#endif
#endif
