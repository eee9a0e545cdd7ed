// sharded_runner.cpp -- the row-partitioned Ell from a plain C++ host (no Python, no collective library of its own):
//   case "group":  EllShardGroup with P = 2 and 3 row blocks on device 0 (in-process exchange) against the unsharded
//                  EllHip on a mixed cut sequence -- xc, kappa, tsq, statuses and every row of Q must be bit-identical
//                  (depth 1: same kernels, same summation shapes);
//   case "driver": the reference's cutting_plane_optim (src/cutting_plane.rs:286-313) on a quadratic oracle with the
//                  group as the search space: same niter / x_best as with EllHip;
//   case "rccl":   ShardedEllHip of ONE rank whose communicator the library creates (ncclCommInitRank, all-gather on
//                  the handle's stream) against EllHip.
// Prints one JSON object per case.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>

#include "../../ellalgo-rs_amd/host/ellhip/sharded_hip.hpp"

using namespace ellhip;

static unsigned long long lcg = 0x9E3779B97F4A7C15ull;
static double urand() {
    lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(lcg >> 11) / 9007199254740992.0;
}
static Arr grad(size_t n) {
    Arr g(n);
    double s = 0.0;
    for (auto& x : g) {
        x = urand() - 0.5;
        s += x * x;
    }
    for (auto& x : g) x /= std::sqrt(s);
    return g;
}
static bool same(const Arr& a, const Arr& b) { return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(double)) == 0; }

struct Quad {  // f(x) = sum (x_i - t_i)^2 (tests/integration_test.rs:85-105)
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

template <class Space>
static void mixed_sequence(Space& a, EllHip& ref, size_t n, int k, bool& ok, int& nsucc) {
    for (int i = 0; i < k; ++i) {
        const Arr g = grad(n);
        const double tau = std::sqrt(ref.tsq() > 0 ? ref.tsq() : 1.0);
        CutStatus sa, sr;
        switch (i % 5) {
            case 0: { std::pair<Arr, SingleCut> c{g, SingleCut{0.1 * tau * urand()}}; sa = a.update_bias_cut(c); sr = ref.update_bias_cut(c); break; }
            case 1: { std::pair<Arr, SingleCut> c{g, SingleCut{0.0}}; sa = a.update_central_cut(c); sr = ref.update_central_cut(c); break; }
            case 2: { std::pair<Arr, ParallelCut> c{g, ParallelCut{0.02 * tau, 0.3 * tau}}; sa = a.update_bias_cut(c); sr = ref.update_bias_cut(c); break; }
            case 3: { std::pair<Arr, ParallelCut> c{g, ParallelCut{0.0, 0.4 * tau}}; sa = a.update_q(c); sr = ref.update_q(c); break; }
            default: { std::pair<Arr, SingleCut> c{g, SingleCut{1e9}}; sa = a.update_bias_cut(c); sr = ref.update_bias_cut(c); break; }  // NoSoln
        }
        ok = ok && sa == sr && a.tsq() == ref.tsq() && a.kappa() == ref.kappa();
        nsucc += sa == CutStatus::Success;
    }
    ok = ok && same(a.xc(), ref.xc());
}

int main() {
    const size_t n = 768;
    for (int P : {2, 3}) {
        EllShardGroup grp = EllShardGroup::new_with_scalar(2.0, Arr(n, 0.25), std::vector<int>((size_t)P, 0));
        EllHip ref = EllHip::new_with_scalar(2.0, Arr(n, 0.25));
        ref.set_defer_depth(1);
        bool ok = true;
        int nsucc = 0;
        mixed_sequence(grp, ref, n, 30, ok, nsucc);
        ok = ok && same(grp.mq(), ref.mq());
        std::printf("{\"case\": \"group%d\", \"ok\": %s, \"successes\": %d, \"blocks\": %zu}\n", P, ok ? "true" : "false", nsucc, grp.nblocks());
    }
    {
        Quad q1, q2;
        q1.target.resize(n);
        for (size_t i = 0; i < n; ++i) q1.target[i] = 0.001 * (double)(i % 17) - 0.005;
        q2 = q1;
        EllShardGroup grp = EllShardGroup::new_with_scalar(10.0, Arr(n, 0.0), {0, 0});
        EllHip ref = EllHip::new_with_scalar(10.0, Arr(n, 0.0));
        ref.set_defer_depth(1);
        double g1 = std::numeric_limits<double>::infinity(), g2 = g1;
        auto [x1, it1] = cutting_plane_optim(q1, grp, g1, Options(60, 1e-12));
        auto [x2, it2] = cutting_plane_optim(q2, ref, g2, Options(60, 1e-12));
        const bool ok = it1 == it2 && g1 == g2 && x1.has_value() == x2.has_value() && (!x1 || same(*x1, *x2));
        std::printf("{\"case\": \"driver\", \"ok\": %s, \"niter\": %zu, \"gamma\": %.17g}\n", ok ? "true" : "false", it1, g1);
    }
    {
        const NcclId id = make_nccl_id();
        ShardedEllHip sh = ShardedEllHip::new_with_scalar(2.0, Arr(n, 0.25), 0, 1, &id);
        EllHip ref = EllHip::new_with_scalar(2.0, Arr(n, 0.25));
        ref.set_defer_depth(1);
        bool ok = true;
        int nsucc = 0;
        mixed_sequence(sh, ref, n, 20, ok, nsucc);
        std::printf("{\"case\": \"rccl\", \"ok\": %s, \"successes\": %d}\n", ok ? "true" : "false", nsucc);
    }
    return 0;
}
