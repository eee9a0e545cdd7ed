// live_loop.cpp -- the number an ellalgo-rs user sees: the reference's hot loop
//     xc() -> oracle -> update_*_cut          (src/cutting_plane.rs:299-311)
// run by the C++ mirrors of its drivers (host/ellhip/cutting_plane.hpp: cutting_plane_optim, and the pipelined form of
// ell_hip.hpp: prime / cut / commit) over the C ABI, with an oracle ON THE HOST that reads the centre it is given (O(n))
// and answers with cut k of a pre-generated sequence -- BASELINE's synthetic cuts, which do not depend on xc, so the
// statuses are those of the queue runs while the data flow is the live one: the gradient of iteration k + 1 does not
// exist before update k has returned.  Iterations per second, fences on both sides, the trailing apply pass inside.
//
// Usage: live_loop <cuts.bin> <warm> <steps> [ell|ellstable]     -> one JSON line on stdout
// cuts.bin (written by bench.py): int64 n, int64 k, int32 kinds[k], double b0[k], double b1[k] (NaN: none), double grads[k][n]
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <optional>
#include <string>
#include <vector>

#include "../ellhip/ell_hip.hpp"

namespace {
using clk = std::chrono::steady_clock;
double us(clk::duration d) { return std::chrono::duration<double, std::micro>(d).count(); }

struct Cuts {
    int64_t n = 0, k = 0;
    std::vector<int32_t> kinds;
    std::vector<double> b0, b1, grads;
};

bool read_cuts(const char* path, Cuts& c) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    bool ok = std::fread(&c.n, 8, 1, f) == 1 && std::fread(&c.k, 8, 1, f) == 1 && c.n > 0 && c.k > 0;
    if (ok) {
        c.kinds.resize((size_t)c.k);
        c.b0.resize((size_t)c.k);
        c.b1.resize((size_t)c.k);
        c.grads.resize((size_t)c.k * (size_t)c.n);
        ok = std::fread(c.kinds.data(), 4, (size_t)c.k, f) == (size_t)c.k && std::fread(c.b0.data(), 8, (size_t)c.k, f) == (size_t)c.k &&
             std::fread(c.b1.data(), 8, (size_t)c.k, f) == (size_t)c.k &&
             std::fread(c.grads.data(), 8, c.grads.size(), f) == c.grads.size();
    }
    std::fclose(f);
    return ok;
}

// OracleOptim (src/cutting_plane.rs:129-136): reads its argument, returns cut number `next` of the sequence; `shrunk`
// = the cut is a central one (the drivers then take x_best = xc and call update_central_cut, :303-306).
struct ReplayOracle {
    const Cuts& c;
    size_t next = 0, warm = 0;
    double sink = 0.0, oracle_us = 0.0;
    clk::time_point t_warm{};
    bool failed = false;
    std::pair<std::pair<ellhip::Arr, ellhip::ParallelCut>, bool> assess_optim(const ellhip::Arr& xc, double& gamma) {
        const clk::time_point t0 = clk::now();
        if (next == warm) {
            t_warm = t0;
            oracle_us = 0.0;
        }
        (void)gamma;
        double acc = 0.0;
        for (const double v : xc) acc += v;   // O(n): the least an oracle does with the centre
        sink += acc;
        const size_t i = next < (size_t)c.k ? next : (size_t)c.k - 1;
        ++next;
        ellhip::Arr g(c.grads.begin() + (ptrdiff_t)(i * (size_t)c.n), c.grads.begin() + (ptrdiff_t)((i + 1) * (size_t)c.n));
        const bool has1 = c.b1[i] == c.b1[i];
        ellhip::ParallelCut cut{c.b0[i], has1 ? std::optional<double>(c.b1[i]) : std::nullopt};
        const bool central = c.kinds[i] == ELLHIP_CUT_CENTRAL;
        oracle_us += us(clk::now() - t0);
        return {{std::move(g), cut}, central};
    }
};

// SearchSpace wrapper that keeps time per call class (plain driver only)
template <class Space>
struct Timed {
    Space& s;
    double xc_us = 0.0, update_us = 0.0;
    long nxc = 0, nupd = 0;
    ellhip::Arr xc() {
        const clk::time_point t0 = clk::now();
        ellhip::Arr x = s.xc();
        xc_us += us(clk::now() - t0);
        ++nxc;
        return x;
    }
    double tsq() const { return s.tsq(); }
    template <class Cut>
    ellhip::CutStatus update_bias_cut(const std::pair<ellhip::Arr, Cut>& cut) {
        const clk::time_point t0 = clk::now();
        const ellhip::CutStatus st = s.update_bias_cut(cut);
        update_us += us(clk::now() - t0);
        ++nupd;
        return st;
    }
    template <class Cut>
    ellhip::CutStatus update_central_cut(const std::pair<ellhip::Arr, Cut>& cut) {
        const clk::time_point t0 = clk::now();
        const ellhip::CutStatus st = s.update_central_cut(cut);
        update_us += us(clk::now() - t0);
        ++nupd;
        return st;
    }
    void reset() { xc_us = update_us = 0.0; nxc = nupd = 0; }
};

template <int VARIANT>
int run(const Cuts& c, size_t warm, size_t steps) {
    using Space = ellhip::SpaceHip<VARIANT>;
    const size_t n = (size_t)c.n;
    const ellhip::Options opt(warm + steps, 0.0);   // tolerance 0: the loop runs its max_iters (tsq < 0 never holds)
    double rate_plain = 0.0, rate_piped = 0.0, xc_us = 0.0, upd_us = 0.0, orc_us = 0.0;
    size_t it_plain = 0, it_piped = 0;
    int depth = 0;
    {   // ---- the reference's loop, statement for statement (cutting_plane_optim)
        Space space = Space::new_with_scalar(1.0, ellhip::Arr(n, 0.0));
        depth = VARIANT == ELLHIP_SPACE_ELL ? space.defer_depth() : 1;
        Timed<Space> ts{space};
        ReplayOracle omega{c};
        omega.warm = warm;
        double gamma = 0.0;
        // (the per-class timers restart when the oracle sees iteration `warm`)
        struct Hook {
            ReplayOracle& o;
            Timed<Space>& t;
            std::pair<std::pair<ellhip::Arr, ellhip::ParallelCut>, bool> assess_optim(const ellhip::Arr& xc, double& g) {
                if (o.next == o.warm) t.reset();
                return o.assess_optim(xc, g);
            }
        } hook{omega, ts};
        auto res = ellhip::cutting_plane_optim(hook, ts, gamma, opt);
        ellhip::check(ellhip_synchronize(space.handle()), "ellhip_synchronize");   // what the last update left in flight
        const double el = us(clk::now() - omega.t_warm);
        it_plain = res.second;
        rate_plain = (double)steps / (el * 1e-6);
        xc_us = ts.xc_us / (double)steps;
        upd_us = ts.update_us / (double)steps;
        orc_us = omega.oracle_us / (double)steps;
    }
    {   // ---- the pipelined drivers (oracle queried between the scalar stage and the shrink)
        Space space = Space::new_with_scalar(1.0, ellhip::Arr(n, 0.0));
        ReplayOracle omega{c};
        omega.warm = warm;
        double gamma = 0.0;
        auto res = ellhip::cutting_plane_optim_pipelined(omega, space, gamma, opt);
        ellhip::check(ellhip_synchronize(space.handle()), "ellhip_synchronize");
        const double el = us(clk::now() - omega.t_warm);
        it_piped = res.second;
        // (the pipelined driver asks the oracle for iteration k + 1 inside iteration k: `warm` oracle calls = warm - 1
        // finished iterations; the clock therefore covers steps + 1 updates and is charged for them)
        rate_piped = (double)(steps + 1) / (el * 1e-6);
    }
    std::printf("{\"case\": \"live_loop\", \"n\": %lld, \"space\": \"%s\", \"defer_depth\": %d, \"warmup\": %zu, \"steps\": %zu, "
                "\"plain_iterations_per_s\": %.2f, \"pipelined_iterations_per_s\": %.2f, \"plain_niter\": %zu, \"pipelined_niter\": %zu, "
                "\"plain_us_per_iteration\": {\"xc\": %.2f, \"oracle\": %.2f, \"update\": %.2f}, "
                "\"driver\": \"host/ellhip/cutting_plane.hpp cutting_plane_optim | ell_hip.hpp cutting_plane_optim_pipelined\", "
                "\"oracle\": \"host, O(n): sums the centre, returns synthetic cut k (a copy of its gradient)\"}\n",
                (long long)c.n, VARIANT == ELLHIP_SPACE_ELL ? "ell" : "ellstable", depth, warm, steps, rate_plain, rate_piped,
                it_plain, it_piped, xc_us, orc_us, upd_us);
    return (it_plain == warm + steps && it_piped == warm + steps) ? 0 : 3;
}
}  // namespace

int main(int argc, char** argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s <cuts.bin> <warm> <steps> [ell|ellstable]\n", argv[0]);
        return 2;
    }
    Cuts c;
    if (!read_cuts(argv[1], c)) {
        std::fprintf(stderr, "cannot read %s\n", argv[1]);
        return 2;
    }
    const size_t warm = (size_t)atoll(argv[2]), steps = (size_t)atoll(argv[3]);
    if ((int64_t)(warm + steps) > c.k) {
        std::fprintf(stderr, "%s holds %lld cuts, %zu asked for\n", argv[1], (long long)c.k, warm + steps);
        return 2;
    }
    try {
        if (argc > 4 && std::string(argv[4]) == "ellstable") return run<ELLHIP_SPACE_ELL_STABLE>(c, warm, steps);
        return run<ELLHIP_SPACE_ELL>(c, warm, steps);
    } catch (const std::exception& e) {
        std::printf("{\"case\": \"live_loop\", \"error\": \"%s\"}\n", e.what());
        return 1;
    }
}
