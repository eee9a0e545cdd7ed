// ellcalc_device.hpp -- EllCalc / EllCalcCore (src/ell_calc.rs) as device code.
//
// The coefficient stage sits BETWEEN the two passes over Q, so it runs on the device (one lane of
// the scalar-stage kernel).  Every expression keeps the reference's operation order; the
// translation unit is compiled with -ffp-contract=off so no a*b+c is fused (rustc never fuses),
// and sqrt / division are the IEEE correctly-rounded device forms.
#pragma once

#include <hip/hip_runtime.h>

namespace ellhip {

enum : int { ST_SUCCESS = 0, ST_NOSOLN = 1, ST_NOEFFECT = 2, ST_UNKNOWN = 3 };
enum : int { CUT_BIAS = 0, CUT_CENTRAL = 1, CUT_Q = 2 };

struct Coef {
    double rho, sigma, delta;
};

// EllCalcCore::new (src/ell_calc.rs:61-78) + EllCalc::new (:653-662)
struct EllCalcDev {
    double n_f, n_plus_1, half_n, inv_n, cst1, cst2;
    int use_parallel_cut;

    __host__ __device__ static EllCalcDev make(long long n, int use_parallel) {
        EllCalcDev c;
        const double n_f = (double)n;
        const double n_sq = n_f * n_f;
        const double cst0 = 1.0 / (n_f + 1.0);
        c.n_f = n_f;
        c.n_plus_1 = n_f + 1.0;
        c.half_n = n_f / 2.0;
        c.inv_n = 1.0 / n_f;
        c.cst1 = n_sq / (n_sq - 1.0);
        c.cst2 = 2.0 * cst0;
        c.use_parallel_cut = use_parallel;
        return c;
    }

    // src/ell_calc.rs:218-240
    __device__ Coef core_parallel_bias_cut_fast(double b0, double b1, double tsq, double b0b1,
                                                double eta) const {
        const double b0sq = b0 * b0;
        const double b1sq = b1 * b1;
        const double zeta0 = tsq - b0sq;
        const double zeta1 = tsq - b1sq;
        const double temp = half_n * (b1sq - b0sq);
        const double xi = sqrt(zeta0 * zeta1 + temp * temp);
        const double bsum = b0 + b1;
        const double bsumsq = bsum * bsum;
        Coef r;
        r.sigma = 2.0 * eta / (tsq + b0b1 + half_n * bsumsq + xi);
        r.rho = r.sigma * (b0 + b1) / 2.0;
        r.delta = cst1 * ((zeta0 + zeta1) / 2.0 + xi / n_f) / tsq;
        return r;
    }
    // src/ell_calc.rs:316-320
    __device__ Coef core_parallel_bias_cut(double b0, double b1, double tsq) const {
        const double b0b1 = b0 * b1;
        const double eta = tsq + n_f * b0b1;
        return core_parallel_bias_cut_fast(b0, b1, tsq, b0b1, eta);
    }
    // src/ell_calc.rs:383-394
    __device__ Coef core_parallel_central_cut(double b1, double tsq) const {
        const double b1sq = b1 * b1;
        const double a1sq = b1sq / tsq;
        const double half_val = half_n * a1sq;
        const double root = half_val + sqrt(1.0 - a1sq + half_val * half_val);
        const double r_plus_1 = root + 1.0;
        Coef r;
        r.rho = b1 / r_plus_1;
        r.sigma = 2.0 / r_plus_1;
        r.delta = root / (root - inv_n);
        return r;
    }
    // src/ell_calc.rs:453-459
    __device__ Coef core_bias_cut_fast(double beta, double tau, double eta) const {
        Coef r;
        r.rho = eta / n_plus_1;
        r.sigma = 2.0 * r.rho / (tau + beta);
        const double alpha = beta / tau;
        r.delta = cst1 * (1.0 - alpha * alpha);
        return r;
    }
    // src/ell_calc.rs:550-553
    __device__ Coef core_bias_cut(double beta, double tau) const {
        const double eta = tau + n_f * beta;
        return core_bias_cut_fast(beta, tau, eta);
    }
    // src/ell_calc.rs:605-611
    __device__ Coef core_central_cut(double tsq) const {
        Coef r;
        r.sigma = cst2;
        r.rho = sqrt(tsq) / n_plus_1;
        r.delta = cst1;
        return r;
    }

    __device__ static int fail(int status, double delta, Coef& out) {
        out.rho = 0.0;
        out.sigma = 0.0;
        out.delta = delta;
        return status;
    }

    // src/ell_calc.rs:870-877
    __device__ int calc_bias_cut(double beta, double tsq, Coef& out) const {
        if (tsq < beta * beta) return fail(ST_NOSOLN, 0.0, out);
        const double tau = sqrt(tsq);
        out = core_bias_cut(beta, tau);
        return ST_SUCCESS;
    }
    // src/ell_calc.rs:892-908
    __device__ int calc_bias_cut_q(double beta, double tsq, Coef& out) const {
        const double tau = sqrt(tsq);
        if (tau < beta) return fail(ST_NOSOLN, 0.0, out);
        const double eta = tau + n_f * beta;
        if (eta < 0.0) return fail(ST_NOEFFECT, 1.0, out);
        out = core_bias_cut_fast(beta, tau, eta);
        return ST_SUCCESS;
    }
    // src/ell_calc.rs:928-931
    __device__ int calc_central_cut(double tsq, Coef& out) const {
        out = core_central_cut(tsq);
        return ST_SUCCESS;
    }
    // src/ell_calc.rs:751-769
    __device__ int calc_parallel_bias_cut(double b0, double b1, double tsq, Coef& out) const {
        if (b1 < b0) return fail(ST_NOSOLN, 0.0, out);
        if ((b1 > 0.0 && tsq <= b1 * b1) || !use_parallel_cut) return calc_bias_cut(b0, tsq, out);
        out = core_parallel_bias_cut(b0, b1, tsq);
        return ST_SUCCESS;
    }
    // src/ell_calc.rs:787-812
    __device__ int calc_parallel_q(double b0, double b1, double tsq, Coef& out) const {
        if (b1 < b0) return fail(ST_NOSOLN, 0.0, out);
        if (((b1 > 0.0) && b1 * b1 >= tsq) || !use_parallel_cut) return calc_bias_cut_q(b0, tsq, out);
        const double b0b1 = b0 * b1;
        const double eta = tsq + n_f * b0b1;
        if (eta <= 0.0) return fail(ST_NOEFFECT, 1.0, out);
        out = core_parallel_bias_cut_fast(b0, b1, tsq, b0b1, eta);
        return ST_SUCCESS;
    }
    // src/ell_calc.rs:836-847
    __device__ int calc_parallel_central_cut(double b1, double tsq, Coef& out) const {
        if (b1 < 0.0) return fail(ST_NOSOLN, 0.0, out);
        if (tsq <= b1 * b1 || !use_parallel_cut) return calc_central_cut(tsq, out);
        out = core_parallel_central_cut(b1, tsq);
        return ST_SUCCESS;
    }

    // CutType dispatch, src/ell.rs:182-210 with src/ell_calc.rs:671-718.
    __device__ int dispatch(int kind, double b0, int has_b1, double b1, double tsq, Coef& out) const {
        switch (kind) {
            case CUT_BIAS:
                return has_b1 ? calc_parallel_bias_cut(b0, b1, tsq, out) : calc_bias_cut(b0, tsq, out);
            case CUT_CENTRAL:
                return has_b1 ? calc_parallel_central_cut(b1, tsq, out) : calc_central_cut(tsq, out);
            case CUT_Q:
                return has_b1 ? calc_parallel_q(b0, b1, tsq, out) : calc_bias_cut_q(b0, tsq, out);
            default:
                return fail(ST_UNKNOWN, 0.0, out);
        }
    }
};

}  // namespace ellhip
