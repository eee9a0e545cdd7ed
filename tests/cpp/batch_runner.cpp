// batch_runner.cpp -- BASELINE config 1 in bulk: many copies of the n = 16 quadratic problem
// (tests/integration_test.rs:85-116 generalised to n = 16) with different targets, solved side by
// side on the batched engine (one cut per ellipsoid per round), and each of them once more on its own
// EllHip handle with the generic cutting_plane_optim driver.  Prints one JSON object per problem.
#include <cmath>
#include <cstdio>
#include <limits>

#include "../../ellalgo-rs_amd/host/ellhip/ell_batch_hip.hpp"

using namespace ellhip;

// f(x) = sum (x_i - t_i)^2, gradient 2 (x - t): the oracle of tests/integration_test.rs:85-105
struct Quad {
    Arr target;
    std::pair<std::pair<Arr, SingleCut>, bool> assess_optim(const Arr& x, double& gamma) {
        double f = 0.0;
        Arr g(x.size());
        for (size_t i = 0; i < x.size(); ++i) {
            const double d = x[i] - target[i];
            f += d * d;
            g[i] = 2.0 * d;
        }
        const double fj = f - gamma;
        if (fj > 0.0) return {{g, SingleCut{fj}}, false};
        gamma = f;
        return {{g, SingleCut{0.0}}, true};
    }
};

int main() {
    const size_t n = 16, B = 24, max_iters = 2000;
    const double tol = 1e-10;
    std::vector<Quad> oracles(B);
    for (size_t b = 0; b < B; ++b) {
        oracles[b].target.resize(n);
        for (size_t i = 0; i < n; ++i) oracles[b].target[i] = (double)(i + 1) + 0.25 * (double)b;
    }
    // ---- one by one (the reference's way): Ell::new_with_scalar(10, 0), Options(2000, 1e-10)
    std::vector<size_t> niter_single(B);
    std::vector<double> gamma_single(B);
    std::vector<Arr> x_single(B);
    for (size_t b = 0; b < B; ++b) {
        EllHip space = EllHip::new_with_scalar(10.0, Arr(n, 0.0));
        double gamma = std::numeric_limits<double>::infinity();
        auto [x, niter] = cutting_plane_optim(oracles[b], space, gamma, Options(max_iters, tol));
        niter_single[b] = niter;
        gamma_single[b] = gamma;
        x_single[b] = x.value_or(Arr(n, 0.0));
    }
    // ---- all together: the same loop with the B spaces in one batch; a finished problem keeps receiving a
    // harmless repeat of its last cut's kind with beta = +inf (NoSoln: state untouched) until all are done
    EllBatchHip batch = EllBatchHip::new_with_scalar(Arr(B, 10.0), std::vector<Arr>(B, Arr(n, 0.0)));
    std::vector<double> gamma(B, std::numeric_limits<double>::infinity());
    std::vector<Arr> x_best(B, Arr(n, 0.0));
    std::vector<size_t> niter(B, max_iters);
    std::vector<bool> done(B, false);
    for (size_t it = 0; it < max_iters; ++it) {
        const std::vector<Arr> xc = batch.xc();
        std::vector<std::pair<Arr, SingleCut>> bias(B), central(B);
        std::vector<bool> shrunk(B, false);
        bool any = false;
        for (size_t b = 0; b < B; ++b) {
            if (done[b]) {
                bias[b] = {Arr(n, 1.0), SingleCut{std::numeric_limits<double>::infinity()}};
                continue;
            }
            any = true;
            auto [cut, sh] = oracles[b].assess_optim(xc[b], gamma[b]);
            shrunk[b] = sh;
            if (sh) x_best[b] = xc[b];
            bias[b] = cut;
        }
        if (!any) break;
        // one launch for the bias cuts, one for the central cuts: an ellipsoid that is not due in a launch gets the
        // no-op cut (beta = +inf -> NoSoln, nothing changes)
        std::vector<std::pair<Arr, SingleCut>> a = bias, c = bias;
        for (size_t b = 0; b < B; ++b) {
            const std::pair<Arr, SingleCut> noop{Arr(n, 1.0), SingleCut{std::numeric_limits<double>::infinity()}};
            if (done[b] || shrunk[b]) a[b] = noop;
            if (done[b] || !shrunk[b]) c[b] = noop;
        }
        // (a failed cut still rewrites tsq, src/ell.rs:105, so tsq is read after the launch that carried the real cut)
        const auto st_a = batch.update_bias_cut(a);
        const Arr tsq_a = batch.tsq();
        const auto st_c = batch.update_central_cut(c);
        const Arr tsq_c = batch.tsq();
        for (size_t b = 0; b < B; ++b) {
            if (done[b]) continue;
            const CutStatus st = shrunk[b] ? st_c[b] : st_a[b];
            const double tsq_b = shrunk[b] ? tsq_c[b] : tsq_a[b];
            if (st != CutStatus::Success || tsq_b < tol) {
                done[b] = true;
                niter[b] = it;
            }
        }
    }
    for (size_t b = 0; b < B; ++b) {
        double dx = 0.0;
        for (size_t i = 0; i < n; ++i) dx = std::fmax(dx, std::fabs(x_best[b][i] - x_single[b][i]));
        printf("{\"case\": \"quad16_%zu\", \"niter_batch\": %zu, \"niter_single\": %zu, \"gamma_batch\": %.17g, "
               "\"gamma_single\": %.17g, \"max_dx\": %.3g, \"x0\": %.17g, \"target0\": %.17g}\n",
               b, niter[b], niter_single[b], gamma[b], gamma_single[b], dx, x_best[b][0], oracles[b].target[0]);
    }
    return 0;
}
