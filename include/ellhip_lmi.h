/*
 * ellhip_lmi.h -- C ABI of the device-side LDLTMgr and LMI oracles (libellhip.so).
 *
 * Reference: `LDLTMgr` (src/oracles/ldlt_mgr.rs:3-141) and the feasibility oracles that drive it,
 * `LMIOracle` (src/oracles/lmi_oracle.rs:5-45: F(x) = B - sum_k x_k F_k must be positive definite) and
 * `LMI0Oracle` (src/oracles/lmi0_oracle.rs:4-35: F(x) = sum_k x_k F_k).  The n matrices F_k (m x m) live in HBM;
 * one call streams them twice (forming F(x), and the quadratic forms of the cut) around an LDL^T with an exit
 * at the first non-positive pivot.  The decision (`pos`), the factor (`storage`) and ep are bit-identical to
 * the reference's arithmetic; the cut gradient agrees to rounding.  Worth it for large blocks (m in the
 * hundreds to thousands); m <= 8192.
 *
 * Same conventions as ellhip.h: host buffers owned by the caller, 0 = ok, negative = ELLHIP_E_*, no CPU fallback.
 */
#ifndef ELLHIP_LMI_H
#define ELLHIP_LMI_H

#include "ellhip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ellhip_lmi ellhip_lmi;

#define ELLHIP_LMI_MMAX 8192

/* LMIOracle::new(mat_f, mat_b) (lmi_oracle.rs:12-21): mat_f = n matrices m*m row-major, contiguous; mat_b m*m.
 * mat_b == NULL: LMI0Oracle::new(mat_f) (lmi0_oracle.rs:10-14).  n == 0 with mat_b: a bare LDLTMgr::new(m) whose
 * ellhip_lmi_assess_feas(x = NULL) is LDLTMgr::factorize(mat_b) (ldlt_mgr.rs:22-24). */
int ellhip_lmi_create(ellhip_lmi **out, int64_t n, int64_t m, const double *mat_f, const double *mat_b, int device);
void ellhip_lmi_destroy(ellhip_lmi *o);

/* OracleFeas::assess_feas(&mut self, xc) -> Option<(Arr, SingleCut)> (lmi_oracle.rs:27-44; lmi0_oracle.rs:16-34).
 * Returns 0 = None (F(x) is positive definite), 1 = Some((g, SingleCut(ep))) with g_out[n] and *ep_out filled. */
int ellhip_lmi_assess_feas(ellhip_lmi *o, const double *x, double *g_out, double *ep_out);

/* LDLTMgr's state after the last call: pos2 = pub pos (start, end) (end == 0: is_spd()). */
int ellhip_lmi_pos(ellhip_lmi *o, int64_t *pos2);
/* pub wit (m): the witness of the last failing factorisation inside [pos.0, pos.1), zero elsewhere. */
int ellhip_lmi_get_witness(ellhip_lmi *o, double *out);
/* the private `storage` (m*m row-major): diagonal = D, strict lower = L, strict upper [k][j] = L[j][k] D[k].
 * Rows at or after a failing pivot hold values the reference never computes. */
int ellhip_lmi_get_storage(ellhip_lmi *o, double *out);
/* LDLTMgr::sqrt (ldlt_mgr.rs:129-140): upper triangular R with A = R'R, m*m row-major; ELLHIP_E_STATE if the
 * last factorisation was not positive definite (the reference asserts). */
int ellhip_lmi_sqrt(ellhip_lmi *o, double *r_out);

#ifdef __cplusplus
}
#endif
#endif
