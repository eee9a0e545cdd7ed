// lmi_runner.cpp -- tests/lmi_tests.rs:115-217 (MyLmiOracle over two LMI oracles, run_lmi / run_lmi_stable) through
// the C++ host mirror with BOTH sides on the MI355X engine: LMIOracleHip + EllHip / EllStableHip.
#include <cstdio>
#include <limits>

#include "../../ellalgo-rs_amd/host/ellhip/lmi_hip.hpp"

using namespace ellhip;

static std::vector<Arr> f1() { return {{-7.0, -11.0, -11.0, 3.0}, {7.0, -18.0, -18.0, 8.0}, {-2.0, -8.0, -8.0, 1.0}}; }
static Arr b1() { return {33.0, -9.0, -9.0, 26.0}; }
static std::vector<Arr> f2() {
    return {{-21.0, -11.0, 0.0, -11.0, 10.0, 8.0, 0.0, 8.0, 5.0},
            {0.0, 10.0, 16.0, 10.0, -10.0, -10.0, 16.0, -10.0, 3.0},
            {-5.0, 2.0, -17.0, 2.0, -6.0, 8.0, -17.0, 8.0, 6.0}};
}
static Arr b2() { return {14.0, 9.0, 40.0, 9.0, 91.0, 10.0, 40.0, 10.0, 15.0}; }

struct MyLmiOracle {  // tests/lmi_tests.rs:121-171
    int idx = -1;
    Arr c{1.0, -1.0, 1.0};
    LMIOracleHip lmi1{f1(), b1(), 2};
    LMIOracleHip lmi2{f2(), b2(), 3};
    std::pair<std::pair<Arr, SingleCut>, bool> assess_optim(const Arr& xc, double& gamma) {
        double f0 = 0.0;
        for (size_t i = 0; i < 3; ++i) f0 += c[i] * xc[i];
        for (int rep = 0; rep < 3; ++rep) {
            idx = (idx == 2) ? 0 : idx + 1;
            if (idx == 0) {
                if (auto cut = lmi1.assess_feas(xc)) return {*cut, false};
            } else if (idx == 1) {
                if (auto cut = lmi2.assess_feas(xc)) return {*cut, false};
            } else {
                const double fj = f0 - gamma;
                if (fj > 0.0) return {{c, SingleCut{fj}}, false};
                gamma = f0;
            }
        }
        return {{c, SingleCut{0.0}}, true};
    }
};

template <class Space>
static void run(const char* name) {
    Space ellip = Space::new_with_scalar(10.0, Arr(3, 0.0));
    MyLmiOracle omega;
    double gamma = std::numeric_limits<double>::infinity();
    auto [x, niter] = cutting_plane_optim(omega, ellip, gamma, Options());
    printf("{\"case\": \"%s\", \"niter\": %zu, \"has_x\": %s, \"gamma\": %.17g, \"x\": [%.17g, %.17g, %.17g]}\n", name, niter,
           x ? "true" : "false", gamma, x ? (*x)[0] : 0.0, x ? (*x)[1] : 0.0, x ? (*x)[2] : 0.0);
}

int main() {
    run<EllHip>("lmi_lazy");
    run<EllStableHip>("lmi_lazy_stable");
    LDLTMgrHip ldlt(3);
    const bool spd = ldlt.factorize({25.0, 15.0, -5.0, 15.0, 18.0, 0.0, -5.0, 0.0, 11.0});
    printf("{\"case\": \"chol1\", \"niter\": 0, \"has_x\": %s, \"gamma\": 0, \"x\": [0, 0, 0]}\n", spd ? "true" : "false");
    return 0;
}
