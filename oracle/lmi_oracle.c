/* lmi_oracle.c -- see lmi_oracle.h.  TEST INFRASTRUCTURE ONLY. */
#include "lmi_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

orc_ldlt *orc_ldlt_new(int64_t ndim) { /* src/oracles/ldlt_mgr.rs:11-18 */
    orc_ldlt *m = (orc_ldlt *)calloc(1, sizeof *m);
    if (!m) return NULL;
    m->ndim = ndim;
    m->wit = (double *)calloc((size_t)ndim, sizeof(double));
    m->storage = (double *)calloc((size_t)(ndim * ndim), sizeof(double));
    return m;
}

void orc_ldlt_free(orc_ldlt *m) {
    if (!m) return;
    free(m->wit);
    free(m->storage);
    free(m);
}

int orc_ldlt_is_spd(const orc_ldlt *m) { return m->pos1 == 0; } /* :93-95 */

static int factor_impl(orc_ldlt *m, orc_elem_fn get, void *ctx, int semidefinite) { /* :29-57, :61-90 */
    const int64_t nd = m->ndim;
    double *st = m->storage;
    int64_t start = 0;
    m->pos0 = 0;
    m->pos1 = 0;
    for (int64_t i = 0; i < nd; ++i) {
        double diag = get(ctx, i, start);
        for (int64_t j = start; j < i; ++j) {
            st[j * nd + i] = diag; /* keep for later */
            const double val = diag / st[j * nd + j];
            st[i * nd + j] = val; /* L[i, j] */
            const int64_t stop = j + 1;
            double s = 0.0;
            for (int64_t k = start; k < stop; ++k) s += st[i * nd + k] * st[k * nd + stop];
            diag = get(ctx, i, stop) - s;
        }
        st[i * nd + i] = diag;
        if (!semidefinite) {
            if (diag <= 0.0) {
                m->pos0 = start;
                m->pos1 = i + 1;
                break;
            }
        } else {
            if (diag < 0.0) {
                m->pos0 = start;
                m->pos1 = i + 1;
                break;
            } else if (diag == 0.0) {
                start = i + 1;
            }
        }
    }
    return orc_ldlt_is_spd(m);
}

int orc_ldlt_factor(orc_ldlt *m, orc_elem_fn get, void *ctx) { return factor_impl(m, get, ctx, 0); }
int orc_ldlt_factor_semidefinite(orc_ldlt *m, orc_elem_fn get, void *ctx) { return factor_impl(m, get, ctx, 1); }

typedef struct {
    const double *mat;
    int64_t nd;
} mat_ctx;
static double mat_elem(void *c, int64_t i, int64_t j) {
    const mat_ctx *mc = (const mat_ctx *)c;
    return mc->mat[i * mc->nd + j];
}
int orc_ldlt_factorize(orc_ldlt *m, const double *mat) { /* :22-24 */
    mat_ctx c = {mat, m->ndim};
    return orc_ldlt_factor(m, mat_elem, &c);
}

double orc_ldlt_witness(orc_ldlt *m) { /* :99-112; caller guarantees !is_spd */
    const int64_t nd = m->ndim, start = m->pos0, pos = m->pos1;
    const int64_t mm = pos - 1;
    m->wit[mm] = 1.0;
    for (int64_t i = mm; i >= start + 1; --i) {
        double s = 0.0;
        for (int64_t k = i; k < pos; ++k) s += m->storage[k * nd + (i - 1)] * m->wit[k];
        m->wit[i - 1] = -s;
    }
    return -m->storage[mm * nd + mm];
}

double orc_ldlt_sym_quad(const orc_ldlt *m, const double *mat) { /* :116-125 */
    const int64_t nd = m->ndim, start = m->pos0, end = m->pos1;
    double result = 0.0;
    for (int64_t i = start; i < end; ++i)
        for (int64_t j = start; j < end; ++j) result += m->wit[i] * mat[i * nd + j] * m->wit[j];
    return result;
}

void orc_ldlt_sqrt(const orc_ldlt *m, double *r) { /* :129-140; caller guarantees is_spd */
    const int64_t nd = m->ndim;
    memset(r, 0, (size_t)(nd * nd) * sizeof(double));
    for (int64_t i = 0; i < nd; ++i) {
        const double val = sqrt(m->storage[i * nd + i]);
        r[i * nd + i] = val;
        for (int64_t j = i + 1; j < nd; ++j) r[i * nd + j] = m->storage[j * nd + i] * val;
    }
}

orc_lmi *orc_lmi_new(int mode, int64_t n, int64_t m, const double *mat_f, const double *mat_b) {
    orc_lmi *o = (orc_lmi *)calloc(1, sizeof *o);
    if (!o) return NULL;
    o->mode = mode;
    o->n = n;
    o->m = m;
    o->mat_f = (double *)malloc((size_t)(n * m * m) * sizeof(double));
    memcpy(o->mat_f, mat_f, (size_t)(n * m * m) * sizeof(double));
    if (mat_b) {
        o->mat_b = (double *)malloc((size_t)(m * m) * sizeof(double));
        memcpy(o->mat_b, mat_b, (size_t)(m * m) * sizeof(double));
    }
    o->ldlt = orc_ldlt_new(m);
    return o;
}

void orc_lmi_free(orc_lmi *o) {
    if (!o) return;
    free(o->mat_f);
    free(o->mat_b);
    orc_ldlt_free(o->ldlt);
    free(o);
}

typedef struct {
    const orc_lmi *o;
    const double *x;
} lmi_ctx;

static double lmi_elem(void *c, int64_t i, int64_t j) {
    const lmi_ctx *lc = (const lmi_ctx *)c;
    const orc_lmi *o = lc->o;
    const int64_t mm = o->m * o->m;
    if (o->mode == 0) { /* src/oracles/lmi_oracle.rs:29-35 */
        double s = o->mat_b[i * o->m + j];
        for (int64_t k = 0; k < o->n; ++k) s -= o->mat_f[k * mm + i * o->m + j] * lc->x[k];
        return s;
    }
    double s = 0.0; /* src/oracles/lmi0_oracle.rs:18-24 */
    for (int64_t k = 0; k < o->n; ++k) s += o->mat_f[k * mm + i * o->m + j] * lc->x[k];
    return s;
}

int orc_lmi_assess_feas(orc_lmi *o, const double *x, double *g, double *ep) {
    lmi_ctx c = {o, x};
    if (orc_ldlt_factor(o->ldlt, lmi_elem, &c)) return 0;
    *ep = orc_ldlt_witness(o->ldlt);
    const int64_t mm = o->m * o->m;
    for (int64_t k = 0; k < o->n; ++k) {
        const double q = orc_ldlt_sym_quad(o->ldlt, o->mat_f + k * mm);
        g[k] = (o->mode == 0) ? q : -q;
    }
    return 1;
}
