/* lowpass_oracle.c -- see lowpass_oracle.h.  TEST INFRASTRUCTURE ONLY. */
#include "lowpass_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PI_F64 3.14159265358979323846264338327950288 /* std::f64::consts::PI */

/* src/oracles/lowpass_oracle.rs:23-53; linspace: src/arr.rs:506-520 */
orc_lowpass *orc_lowpass_new(int64_t ndim, double wpass, double wstop, double lp_sq, double up_sq,
                             double sp_sq) {
    orc_lowpass *o = (orc_lowpass *)calloc(1, sizeof *o);
    if (!o) return NULL;
    const int64_t mdim = 15 * ndim;
    o->ndim = ndim;
    o->mdim = mdim;
    o->spectrum = (double *)malloc((size_t)mdim * (size_t)ndim * sizeof(double));
    if (!o->spectrum) {
        free(o);
        return NULL;
    }
    const double step = (mdim > 1) ? (PI_F64 - 0.0) / (double)(mdim - 1) : 0.0;
    for (int64_t i = 0; i < mdim; ++i) {
        const double omega = (mdim > 1) ? 0.0 + step * (double)i : 0.0;
        double *row = o->spectrum + i * ndim;
        row[0] = 1.0;
        for (int64_t j = 1; j < ndim; ++j) row[j] = 2.0 * cos(omega * (double)j);
    }
    o->nwpass = (int)floor(wpass * (double)(mdim - 1)) + 1;
    o->nwstop = (int)floor(wstop * (double)(mdim - 1)) + 1;
    o->more_alt = 1;
    o->idx1 = -1;
    o->lp_sq = lp_sq;
    o->up_sq = up_sq;
    o->sp_sq = sp_sq;
    o->idx2 = o->nwpass - 1;
    o->idx3 = o->nwstop - 1;
    o->fmax = -INFINITY;
    o->kmax = -1;
    return o;
}

void orc_lowpass_free(orc_lowpass *o) {
    if (!o) return;
    free(o->spectrum);
    free(o);
}

/* :153-167 */
void orc_lowpass_case(int corrected, double out5[5]) {
    const double delta0_wpass = 0.025;
    const double delta0_wstop = 0.125;
    double delta1, delta2;
    if (corrected) {
        delta1 = 20.0 * log10(1.0 + delta0_wpass);
        delta2 = 20.0 * log10(delta0_wstop);
    } else {
        delta1 = 20.0 * log10(delta0_wpass * PI_F64);
        delta2 = 20.0 * log10(delta0_wstop * PI_F64);
    }
    const double low_pass = pow(10.0, -delta1 / 20.0);
    const double up_pass = pow(10.0, delta1 / 20.0);
    const double stop_pass = pow(10.0, delta2 / 20.0);
    out5[0] = 0.12;
    out5[1] = 0.20;
    out5[2] = low_pass * low_pass;
    out5[3] = up_pass * up_pass;
    out5[4] = stop_pass * stop_pass;
}

/* Arr::dot, src/arr.rs:443-451: left fold from 0.0 */
static double row_dot(const double *a, const double *x, int64_t n) {
    double s = 0.0;
    for (int64_t j = 0; j < n; ++j) s += a[j] * x[j];
    return s;
}

static void put_row(double *g, const double *row, int64_t n, int negate) {
    if (negate)
        for (int64_t j = 0; j < n; ++j) g[j] = -row[j];
    else
        memcpy(g, row, (size_t)n * sizeof(double));
}

/* :58-133 */
int orc_lowpass_assess_feas(orc_lowpass *o, const double *x, double *g, double *b0, int *has_b1,
                            double *b1) {
    o->more_alt = 1;
    o->rows_visited = 0;
    const int mdim = (int)o->mdim;
    const int64_t ndim = o->ndim;
    for (int t = 0; t < o->nwpass; ++t) {
        o->idx1 += 1;
        if (o->idx1 == o->nwpass) o->idx1 = 0;
        const double *col_k = o->spectrum + (int64_t)o->idx1 * ndim;
        const double val = row_dot(col_k, x, ndim);
        o->rows_visited++;
        if (val > o->up_sq) {
            *b0 = val - o->up_sq;
            *has_b1 = 1;
            *b1 = val - o->lp_sq;
            put_row(g, col_k, ndim, 0);
            return 1;
        }
        if (val < o->lp_sq) {
            *b0 = -val + o->lp_sq;
            *has_b1 = 1;
            *b1 = -val + o->up_sq;
            put_row(g, col_k, ndim, 1);
            return 1;
        }
    }
    o->fmax = -INFINITY;
    o->kmax = -1;
    for (int t = o->nwstop; t < mdim; ++t) {
        o->idx3 += 1;
        if (o->idx3 == mdim) o->idx3 = o->nwstop;
        const double *col_k = o->spectrum + (int64_t)o->idx3 * ndim;
        const double val = row_dot(col_k, x, ndim);
        o->rows_visited++;
        if (val > o->sp_sq) {
            *b0 = val - o->sp_sq;
            *has_b1 = 1;
            *b1 = val;
            put_row(g, col_k, ndim, 0);
            return 1;
        }
        if (val < 0.0) {
            *b0 = -val;
            *has_b1 = 1;
            *b1 = -val + o->sp_sq;
            put_row(g, col_k, ndim, 1);
            return 1;
        }
        if (val > o->fmax) {
            o->fmax = val;
            o->kmax = o->idx3;
        }
    }
    for (int t = o->nwpass; t < o->nwstop; ++t) {
        o->idx2 += 1;
        if (o->idx2 == o->nwstop) o->idx2 = o->nwpass;
        const double *col_k = o->spectrum + (int64_t)o->idx2 * ndim;
        const double val = row_dot(col_k, x, ndim);
        o->rows_visited++;
        if (val < 0.0) {
            *b0 = -val;
            *has_b1 = 0;
            *b1 = 0.0;
            put_row(g, col_k, ndim, 1);
            return 1;
        }
    }
    o->more_alt = 0;
    if (x[0] < 0.0) {
        memset(g, 0, (size_t)ndim * sizeof(double));
        g[0] = -1.0;
        *b0 = -x[0];
        *has_b1 = 0;
        *b1 = 0.0;
        return 1;
    }
    return 0;
}

/* :139-150 */
int orc_lowpass_assess_optim(orc_lowpass *o, const double *x, double *gamma, double *g, double *b0,
                             int *has_b1, double *b1, int *shrunk) {
    o->sp_sq = *gamma;
    if (orc_lowpass_assess_feas(o, x, g, b0, has_b1, b1)) {
        *shrunk = 0;
        return 1;
    }
    if (o->kmax < 0) return -1;
    put_row(g, o->spectrum + (int64_t)o->kmax * o->ndim, o->ndim, 0);
    *b0 = 0.0;
    *has_b1 = 1;
    *b1 = o->fmax;
    *gamma = o->fmax;
    *shrunk = 1;
    return 1;
}

static const double *space_xc(int kind, void *sp) {
    return kind == 0 ? orc_ell_xc((orc_ell *)sp) : orc_ellstable_xc((orc_ellstable *)sp);
}
static double space_tsq(int kind, void *sp) {
    return kind == 0 ? orc_ell_tsq((orc_ell *)sp) : orc_ellstable_tsq((orc_ellstable *)sp);
}
static int space_update(int kind, void *sp, int cut, const double *g, double b0, int has_b1, double b1) {
    return kind == 0 ? orc_ell_update((orc_ell *)sp, cut, g, b0, has_b1, b1)
                     : orc_ellstable_update((orc_ellstable *)sp, cut, g, b0, has_b1, b1);
}

/* src/cutting_plane.rs:286-313 */
int64_t orc_lowpass_cutting_plane_optim(orc_lowpass *o, int space_kind, void *space, double *gamma,
                                        int64_t max_iters, double tol, double *x_best, int *has_best,
                                        int *last_status) {
    const int64_t n = o->ndim;
    double *g = (double *)malloc((size_t)n * sizeof(double));
    double *x = (double *)malloc((size_t)n * sizeof(double));
    *has_best = 0;
    *last_status = ORC_SUCCESS;
    int64_t niter;
    for (niter = 0; niter < max_iters; ++niter) {
        double b0, b1;
        int has_b1, shrunk;
        memcpy(x, space_xc(space_kind, space), (size_t)n * sizeof(double)); /* &space.xc() clones */
        if (orc_lowpass_assess_optim(o, x, gamma, g, &b0, &has_b1, &b1, &shrunk) < 0) {
            *last_status = ORC_UNKNOWN;
            break;
        }
        int status;
        if (shrunk) {
            memcpy(x_best, x, (size_t)n * sizeof(double));
            *has_best = 1;
            status = space_update(space_kind, space, ORC_CUT_CENTRAL, g, b0, has_b1, b1);
        } else {
            status = space_update(space_kind, space, ORC_CUT_BIAS, g, b0, has_b1, b1);
        }
        *last_status = status;
        if (status != ORC_SUCCESS || space_tsq(space_kind, space) < tol) break;
    }
    free(g);
    free(x);
    return niter;
}

/* src/cutting_plane.rs:205-227 */
int64_t orc_lowpass_cutting_plane_feas(orc_lowpass *o, int space_kind, void *space, int64_t max_iters,
                                       double tol, double *x_out, int *feasible, int *last_status) {
    const int64_t n = o->ndim;
    double *g = (double *)malloc((size_t)n * sizeof(double));
    double *x = (double *)malloc((size_t)n * sizeof(double));
    *feasible = 0;
    *last_status = ORC_SUCCESS;
    int64_t niter;
    for (niter = 0; niter < max_iters; ++niter) {
        double b0, b1;
        int has_b1;
        memcpy(x, space_xc(space_kind, space), (size_t)n * sizeof(double));
        if (!orc_lowpass_assess_feas(o, x, g, &b0, &has_b1, &b1)) {
            memcpy(x_out, x, (size_t)n * sizeof(double));
            *feasible = 1;
            break;
        }
        const int status = space_update(space_kind, space, ORC_CUT_BIAS, g, b0, has_b1, b1);
        *last_status = status;
        if (status != ORC_SUCCESS || space_tsq(space_kind, space) < tol) break;
    }
    free(g);
    free(x);
    return niter;
}
