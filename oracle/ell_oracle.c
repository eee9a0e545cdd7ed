/*
 * ell_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, see ell_oracle.h).
 *
 * Plain-C restatement of the reference arithmetic; every function cites the reference lines it
 * follows (paths relative to the ellalgo-rs 0.1.7 tree).  Operation order is kept exactly: no
 * re-association, no FMA (compile with -ffp-contract=off), sums are left folds.
 */
#include "ell_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- EllCalcCore / EllCalc ---- */

/* src/ell_calc.rs:61-78 (EllCalcCore::new) and :653-662 (EllCalc::new). */
void orc_calc_init(orc_calc *c, int64_t n) {
    double n_f = (double)n;
    double n_sq = n_f * n_f;
    double cst0 = 1.0 / (n_f + 1.0);
    c->n_f = n_f;
    c->n_plus_1 = n_f + 1.0;
    c->half_n = n_f / 2.0;
    c->inv_n = 1.0 / n_f;
    c->cst1 = n_sq / (n_sq - 1.0);
    c->cst2 = 2.0 * cst0;
    c->use_parallel_cut = 1;
}

/* src/ell_calc.rs:218-240. */
void orc_core_parallel_bias_cut_fast(const orc_calc *c, double b0, double b1, double tsq,
                                     double b0b1, double eta, double out[3]) {
    double b0sq = b0 * b0;
    double b1sq = b1 * b1;
    double zeta0 = tsq - b0sq;
    double zeta1 = tsq - b1sq;
    double temp = c->half_n * (b1sq - b0sq);
    double xi = sqrt(zeta0 * zeta1 + temp * temp);
    double bsum = b0 + b1;
    double bsumsq = bsum * bsum;
    double sigma = 2.0 * eta / (tsq + b0b1 + c->half_n * bsumsq + xi);
    double rho = sigma * (b0 + b1) / 2.0;
    double delta = c->cst1 * ((zeta0 + zeta1) / 2.0 + xi / c->n_f) / tsq;
    out[0] = rho;
    out[1] = sigma;
    out[2] = delta;
}

/* src/ell_calc.rs:316-320. */
void orc_core_parallel_bias_cut(const orc_calc *c, double b0, double b1, double tsq, double out[3]) {
    double b0b1 = b0 * b1;
    double eta = tsq + c->n_f * b0b1;
    orc_core_parallel_bias_cut_fast(c, b0, b1, tsq, b0b1, eta, out);
}

/* src/ell_calc.rs:383-394. */
void orc_core_parallel_central_cut(const orc_calc *c, double b1, double tsq, double out[3]) {
    double b1sq = b1 * b1;
    double a1sq = b1sq / tsq;
    double half_val = c->half_n * a1sq;
    double root = half_val + sqrt(1.0 - a1sq + half_val * half_val);
    double r_plus_1 = root + 1.0;
    out[0] = b1 / r_plus_1;
    out[1] = 2.0 / r_plus_1;
    out[2] = root / (root - c->inv_n);
}

/* src/ell_calc.rs:453-459. */
void orc_core_bias_cut_fast(const orc_calc *c, double beta, double tau, double eta, double out[3]) {
    double rho = eta / c->n_plus_1;
    double sigma = 2.0 * rho / (tau + beta);
    double alpha = beta / tau;
    double delta = c->cst1 * (1.0 - alpha * alpha);
    out[0] = rho;
    out[1] = sigma;
    out[2] = delta;
}

/* src/ell_calc.rs:550-553. */
void orc_core_bias_cut(const orc_calc *c, double beta, double tau, double out[3]) {
    double eta = tau + c->n_f * beta;
    orc_core_bias_cut_fast(c, beta, tau, eta, out);
}

/* src/ell_calc.rs:605-611. */
void orc_core_central_cut(const orc_calc *c, double tsq, double out[3]) {
    out[1] = c->cst2;
    out[0] = sqrt(tsq) / c->n_plus_1;
    out[2] = c->cst1;
}

static int fail3(int status, double d, double out[3]) {
    out[0] = 0.0;
    out[1] = 0.0;
    out[2] = d;
    return status;
}

/* src/ell_calc.rs:870-877. */
int orc_calc_bias_cut(const orc_calc *c, double beta, double tsq, double out[3]) {
    if (tsq < beta * beta) return fail3(ORC_NOSOLN, 0.0, out);
    double tau = sqrt(tsq);
    orc_core_bias_cut(c, beta, tau, out);
    return ORC_SUCCESS;
}

/* src/ell_calc.rs:892-908. */
int orc_calc_bias_cut_q(const orc_calc *c, double beta, double tsq, double out[3]) {
    double tau = sqrt(tsq);
    if (tau < beta) return fail3(ORC_NOSOLN, 0.0, out);
    double eta = tau + c->n_f * beta;
    if (eta < 0.0) return fail3(ORC_NOEFFECT, 1.0, out);
    orc_core_bias_cut_fast(c, beta, tau, eta, out);
    return ORC_SUCCESS;
}

/* src/ell_calc.rs:928-931. */
int orc_calc_central_cut(const orc_calc *c, double tsq, double out[3]) {
    orc_core_central_cut(c, tsq, out);
    return ORC_SUCCESS;
}

/* src/ell_calc.rs:751-769. */
int orc_calc_parallel_bias_cut(const orc_calc *c, double b0, double b1, double tsq, double out[3]) {
    if (b1 < b0) return fail3(ORC_NOSOLN, 0.0, out);
    if ((b1 > 0.0 && tsq <= b1 * b1) || !c->use_parallel_cut) return orc_calc_bias_cut(c, b0, tsq, out);
    orc_core_parallel_bias_cut(c, b0, b1, tsq, out);
    return ORC_SUCCESS;
}

/* src/ell_calc.rs:787-812. */
int orc_calc_parallel_q(const orc_calc *c, double b0, double b1, double tsq, double out[3]) {
    if (b1 < b0) return fail3(ORC_NOSOLN, 0.0, out);
    if (((b1 > 0.0) && b1 * b1 >= tsq) || !c->use_parallel_cut) return orc_calc_bias_cut_q(c, b0, tsq, out);
    double b0b1 = b0 * b1;
    double eta = tsq + c->n_f * b0b1;
    if (eta <= 0.0) return fail3(ORC_NOEFFECT, 1.0, out);
    orc_core_parallel_bias_cut_fast(c, b0, b1, tsq, b0b1, eta, out);
    return ORC_SUCCESS;
}

/* src/ell_calc.rs:836-847. */
int orc_calc_parallel_central_cut(const orc_calc *c, double b1, double tsq, double out[3]) {
    if (b1 < 0.0) return fail3(ORC_NOSOLN, 0.0, out);
    if (tsq <= b1 * b1 || !c->use_parallel_cut) return orc_calc_central_cut(c, tsq, out);
    orc_core_parallel_central_cut(c, b1, tsq, out);
    return ORC_SUCCESS;
}

/* src/ell.rs:182-210 (CutType for SingleCut / ParallelCut) + src/ell_calc.rs:671-718. */
int orc_calc_dispatch(const orc_calc *c, int kind, double b0, int has_b1, double b1, double tsq,
                      double out[3]) {
    switch (kind) {
    case ORC_CUT_BIAS:
        return has_b1 ? orc_calc_parallel_bias_cut(c, b0, b1, tsq, out) : orc_calc_bias_cut(c, b0, tsq, out);
    case ORC_CUT_CENTRAL:
        return has_b1 ? orc_calc_parallel_central_cut(c, b1, tsq, out) : orc_calc_central_cut(c, tsq, out);
    case ORC_CUT_Q:
        return has_b1 ? orc_calc_parallel_q(c, b0, b1, tsq, out) : orc_calc_bias_cut_q(c, b0, tsq, out);
    default:
        return fail3(ORC_UNKNOWN, 0.0, out);
    }
}

/* ------------------------------------------------------------------------------- Ell ------- */

static double *dup_or_build(int64_t n, const double *mq, const double *diag) {
    double *m = (double *)calloc((size_t)n * (size_t)n, sizeof(double));
    if (!m) return NULL;
    if (mq) {
        memcpy(m, mq, (size_t)n * (size_t)n * sizeof(double));
    } else {
        /* Arr::eye / Arr::from_diag, src/arr.rs:40-55 */
        for (int64_t i = 0; i < n; ++i) m[i * n + i] = diag ? diag[i] : 1.0;
    }
    return m;
}

/* src/ell.rs:31-78 (constructors). */
orc_ell *orc_ell_new(int64_t n, double kappa, const double *mq, const double *diag, const double *xc) {
    orc_ell *e = (orc_ell *)calloc(1, sizeof(orc_ell));
    e->n = n;
    e->mq = dup_or_build(n, mq, diag);
    e->xc = (double *)calloc((size_t)n, sizeof(double));
    e->gt = (double *)calloc((size_t)n, sizeof(double));
    if (xc) memcpy(e->xc, xc, (size_t)n * sizeof(double));
    e->kappa = kappa;
    e->tsq = 0.0;
    e->no_defer_trick = 0;
    orc_calc_init(&e->helper, n);
    return e;
}

orc_ell *orc_ell_clone(const orc_ell *s) {
    orc_ell *e = orc_ell_new(s->n, s->kappa, s->mq, NULL, s->xc);
    e->tsq = s->tsq;
    e->no_defer_trick = s->no_defer_trick;
    e->helper = s->helper;
    return e;
}

void orc_ell_free(orc_ell *e) {
    if (!e) return;
    free(e->mq);
    free(e->xc);
    free(e->gt);
    free(e);
}

/* Arr::dot_mv, src/arr.rs:426-442: per row, acc starts at 0.0, j ascending. */
void orc_rows_gemv(int64_t n, int64_t row0, int64_t nrows, const double *mq_local,
                   const double *grad, double *gt_full) {
    for (int64_t i = 0; i < nrows; ++i) {
        const double *row = mq_local + i * n;
        double acc = 0.0;
        for (int64_t j = 0; j < n; ++j) acc += row[j] * grad[j];
        gt_full[row0 + i] = acc;
    }
}

/* Arr::dot, src/arr.rs:443-451: left fold of the products. */
static double dot_fold(int64_t n, const double *a, const double *b) {
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* Ell::update_core, src/ell.rs:97-137.  rowwise selects how the symmetric rank-1 is applied. */
static int ell_update_impl(orc_ell *e, int kind, const double *grad, double b0, int has_b1,
                           double b1, int rowwise) {
    const int64_t n = e->n;
    double *gt = e->gt;
    double *mq = e->mq;
    double coef[3];

    orc_rows_gemv(n, 0, n, mq, grad, gt);            /* :102 */
    double omega = dot_fold(n, grad, gt);             /* :103 */
    e->tsq = e->kappa * omega;                        /* :105 */
    int status = orc_calc_dispatch(&e->helper, kind, b0, has_b1, b1, e->tsq, coef); /* :106 */
    if (status != ORC_SUCCESS) return status;         /* :107-109 */
    double rho = coef[0], sigma = coef[1], delta = coef[2];

    double rho_over_omega = rho / omega;              /* :112 */
    for (int64_t i = 0; i < n; ++i) e->xc[i] -= rho_over_omega * gt[i]; /* :113-115 */

    double ratio = sigma / omega;                     /* :117 */
    if (!rowwise) {
        for (int64_t i = 0; i < n; ++i) {             /* :118-128 */
            double r_qg = ratio * gt[i];
            for (int64_t j = 0; j <= i; ++j) {
                double update = r_qg * gt[j];
                int64_t idx = i * n + j;
                mq[idx] -= update;
                if (i != j) mq[j * n + i] = mq[idx];
            }
        }
    } else {
        for (int64_t i = 0; i < n; ++i) {
            double *row = mq + i * n;
            double gti = gt[i];
            double r_i = ratio * gti;
            for (int64_t j = 0; j <= i; ++j) row[j] -= r_i * gt[j];
            for (int64_t j = i + 1; j < n; ++j) row[j] -= (ratio * gt[j]) * gti;
        }
    }

    e->kappa *= delta;                                /* :130 */
    if (e->no_defer_trick) {                          /* :132-135, MulAssign<f64> src/arr.rs:233-240 */
        double k = e->kappa;
        for (int64_t t = 0; t < n * n; ++t) mq[t] *= k;
        e->kappa = 1.0;
    }
    return status;
}

int orc_ell_update(orc_ell *e, int kind, const double *grad, double b0, int has_b1, double b1) {
    return ell_update_impl(e, kind, grad, b0, has_b1, b1, 0);
}

int orc_ell_update_rowwise(orc_ell *e, int kind, const double *grad, double b0, int has_b1, double b1) {
    return ell_update_impl(e, kind, grad, b0, has_b1, b1, 1);
}

/* NOT the reference's loop: the same arithmetic with the rows of the GEMV and of the row-wise rank-1 spread over
 * OpenMP threads (each row's sum is still a left fold; omega is folded serially), for the "all cores" line of
 * bench.py's cpu_baseline.  Bit-identical to orc_ell_update_rowwise.  Symmetric Q only. */
#ifdef _OPENMP
#include <omp.h>
#endif
int orc_set_num_threads(int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    return omp_get_max_threads();
#else
    (void)nthreads;
    return 1;
#endif
}

int orc_ell_update_rowwise_mt(orc_ell *e, int kind, const double *grad, double b0, int has_b1, double b1) {
    const int64_t n = e->n;
    double *gt = e->gt;
    double *mq = e->mq;
    double coef[3];
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const double *row = mq + i * n;
        double acc = 0.0;
        for (int64_t j = 0; j < n; ++j) acc += row[j] * grad[j];
        gt[i] = acc;
    }
    double omega = dot_fold(n, grad, gt);
    e->tsq = e->kappa * omega;
    int status = orc_calc_dispatch(&e->helper, kind, b0, has_b1, b1, e->tsq, coef);
    if (status != ORC_SUCCESS) return status;
    const double rho_over_omega = coef[0] / omega;
    for (int64_t i = 0; i < n; ++i) e->xc[i] -= rho_over_omega * gt[i];
    const double ratio = coef[1] / omega;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double *row = mq + i * n;
        const double gti = gt[i];
        const double r_i = ratio * gti;
        for (int64_t j = 0; j <= i; ++j) row[j] -= r_i * gt[j];
        for (int64_t j = i + 1; j < n; ++j) row[j] -= (ratio * gt[j]) * gti;
    }
    e->kappa *= coef[2];
    if (e->no_defer_trick) {
        const double k = e->kappa;
#pragma omp parallel for schedule(static)
        for (int64_t t = 0; t < n * n; ++t) mq[t] *= k;
        e->kappa = 1.0;
    }
    return status;
}

double orc_ell_kappa(const orc_ell *e) { return e->kappa; }
double orc_ell_tsq(const orc_ell *e) { return e->tsq; }
double *orc_ell_mq(orc_ell *e) { return e->mq; }
double *orc_ell_xc(orc_ell *e) { return e->xc; }
void orc_ell_set_no_defer_trick(orc_ell *e, int flag) { e->no_defer_trick = flag; }
void orc_ell_set_use_parallel_cut(orc_ell *e, int flag) { e->helper.use_parallel_cut = flag; }

/* ------------------------------------------------------------------------- EllStable ------- */

/* src/ell_stable.rs:18-35 (constructors). */
orc_ellstable *orc_ellstable_new(int64_t n, double kappa, const double *mq, const double *diag,
                                 const double *xc) {
    orc_ellstable *e = (orc_ellstable *)calloc(1, sizeof(orc_ellstable));
    e->n = n;
    e->mq = dup_or_build(n, mq, diag);
    e->xc = (double *)calloc((size_t)n, sizeof(double));
    if (xc) memcpy(e->xc, xc, (size_t)n * sizeof(double));
    e->w = (double *)calloc((size_t)n, sizeof(double));
    e->z = (double *)calloc((size_t)n, sizeof(double));
    e->gg = (double *)calloc((size_t)n, sizeof(double));
    e->q = (double *)calloc((size_t)n, sizeof(double));
    e->kappa = kappa;
    e->tsq = 0.0;
    e->corrected = 0;
    orc_calc_init(&e->helper, n);
    return e;
}

orc_ellstable *orc_ellstable_clone(const orc_ellstable *s) {
    orc_ellstable *e = orc_ellstable_new(s->n, s->kappa, s->mq, NULL, s->xc);
    e->tsq = s->tsq;
    e->corrected = s->corrected;
    e->helper = s->helper;
    return e;
}

void orc_ellstable_free(orc_ellstable *e) {
    if (!e) return;
    free(e->mq);
    free(e->xc);
    free(e->w);
    free(e->z);
    free(e->gg);
    free(e->q);
    free(e);
}

/* EllStable::update_core, src/ell_stable.rs:52-125. */
int orc_ellstable_update(orc_ellstable *e, int kind, const double *grad, double b0, int has_b1,
                         double b1) {
    const int64_t n = e->n;
    double *mq = e->mq;
    double *w = e->w, *z = e->z, *gg = e->gg, *q = e->q;
    double coef[3];

    /* :61-69  w = inv(L) g; the products are parked in the strict lower triangle */
    memcpy(w, grad, (size_t)n * sizeof(double));
    for (int64_t i = 1; i < n; ++i) {
        for (int64_t j = 0; j < i; ++j) {
            double val = mq[j * n + i] * w[j];
            mq[i * n + j] = val;
            w[i] -= val;
        }
    }
    /* :72-75 */
    for (int64_t i = 0; i < n; ++i) z[i] = w[i] * mq[i * n + i];
    /* :78-83 */
    double omega = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        gg[i] = z[i] * w[i];
        omega += gg[i];
    }
    e->tsq = e->kappa * omega;                        /* :85 */
    int status = orc_calc_dispatch(&e->helper, kind, b0, has_b1, b1, e->tsq, coef); /* :86 */
    if (status != ORC_SUCCESS) return status;         /* :88-90 */
    double rho = coef[0], sigma = coef[1], delta = coef[2];

    /* :93-98  back substitution; the reference reads the parked products mq.at(j, i-1), j >= i */
    memcpy(q, z, (size_t)n * sizeof(double));
    for (int64_t i = n - 1; i >= 1; --i) {
        for (int64_t j = i; j < n; ++j) {
            double lij = e->corrected ? mq[(i - 1) * n + j] : mq[j * n + (i - 1)];
            q[i - 1] -= lij * q[j];
        }
    }
    /* :101-104 */
    double rho_over_omega = rho / omega;
    for (int64_t i = 0; i < n; ++i) e->xc[i] -= rho_over_omega * q[i];

    /* :107-121  rank-one update of the factor */
    double mu_val = sigma / (1.0 - sigma);
    double oldt = omega / mu_val;
    const int64_t last = n - 1;
    if (!e->corrected) {
        for (int64_t j = 0; j < last; ++j) {
            double temp = oldt + gg[j];
            double beta2 = z[j] / temp;
            mq[j * n + j] *= oldt / temp;
            for (int64_t l = j + 1; l < n; ++l) mq[j * n + l] += beta2 * mq[l * n + j];
            oldt = temp;
        }
    } else {
        /* consistent variant: v = partially eliminated g (v_l = g_l - sum_{k<=j} parked[l][k]) */
        double *v = q; /* q no longer needed */
        memcpy(v, grad, (size_t)n * sizeof(double));
        for (int64_t j = 0; j < last; ++j) {
            double temp = oldt + gg[j];
            double beta2 = z[j] / temp;
            mq[j * n + j] *= oldt / temp;
            for (int64_t l = j + 1; l < n; ++l) {
                v[l] -= mq[l * n + j];
                mq[j * n + l] += beta2 * v[l];
            }
            oldt = temp;
        }
    }
    {
        double temp = oldt + gg[last];
        mq[last * n + last] *= oldt / temp;
    }
    e->kappa *= delta;                                /* :122 */
    return status;
}

double orc_ellstable_kappa(const orc_ellstable *e) { return e->kappa; }
double orc_ellstable_tsq(const orc_ellstable *e) { return e->tsq; }
double *orc_ellstable_mq(orc_ellstable *e) { return e->mq; }
double *orc_ellstable_xc(orc_ellstable *e) { return e->xc; }
void orc_ellstable_set_corrected(orc_ellstable *e, int flag) { e->corrected = flag; }

/* B independent Ell spaces (identity, kappa0, xc = 0), K cuts each, as a plain loop over orc_ell_update:
 * the CPU statement of what the batched engine (include/ellhip_batch.h) does.  kinds/b0/has_b1/b1 are [K][B],
 * grads [K][B][n]; outputs may be NULL.  Returns the number of Success cuts. */
int64_t orc_ell_batch_run(int64_t B, int64_t n, int64_t K, const int32_t *kinds, const double *grads,
                          const double *b0, const int32_t *has_b1, const double *b1, double kappa0,
                          int32_t *status_out, double *mq_out, double *xc_out, double *kappa_out) {
    int64_t ok = 0;
    for (int64_t b = 0; b < B; ++b) {
        orc_ell *e = orc_ell_new(n, kappa0, NULL, NULL, NULL);
        if (!e) return -1;
        for (int64_t k = 0; k < K; ++k) {
            const int64_t c = k * B + b;
            const int st = orc_ell_update(e, kinds[c], grads + c * n, b0[c], has_b1[c], b1[c]);
            if (status_out) status_out[c] = st;
            ok += st == ORC_SUCCESS;
        }
        if (mq_out) memcpy(mq_out + b * n * n, e->mq, (size_t)(n * n) * sizeof(double));
        if (xc_out) memcpy(xc_out + b * n, e->xc, (size_t)n * sizeof(double));
        if (kappa_out) kappa_out[b] = e->kappa;
        orc_ell_free(e);
    }
    return ok;
}
