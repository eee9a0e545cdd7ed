"""The reference's end-to-end known answers: iteration counts pinned by
`assert_eq!` in the reference's own tests, plus the weaker assertions next to them."""
import math

# case -> (niter or None, has_x or None, flag or None, source)
PINNED = {
    "example1_feasible": (25, True, None, "src/example1.rs:47-49"),
    "example1_infeasible1": (None, False, None, "src/example1.rs:60"),
    "example1_infeasible2": (None, False, None, "src/example1.rs:69"),
    "example1_rr_feasible": (25, True, None, "src/example1_rr.rs:72-73"),
    "example1_rr_infeasible1": (None, False, None, "src/example1_rr.rs:84"),
    "example1_rr_infeasible2": (None, False, None, "src/example1_rr.rs:93"),
    "example4_feasible": (82, True, None, "src/example4.rs:75-76"),
    "quasicvx_feasible": (35, True, None, "src/quasicvx.rs:72-77"),
    "quasicvx_infeasible1": (None, False, None, "src/quasicvx.rs:87"),
    "quasicvx_infeasible2": (None, False, None, "src/quasicvx.rs:96"),
    "quasicvx_feasible_stable": (None, True, None, "src/quasicvx.rs:110"),
    "quasicvx_infeasible1_stable": (None, False, None, "src/quasicvx.rs:120"),
    "quasicvx_infeasible2_stable": (None, False, None, "src/quasicvx.rs:129"),
    "example3_bsearch": (34, None, 1, "src/example3.rs:82-84"),
    "profit": (83, True, None, "src/oracles/profit_oracle.rs:183-187"),
    "profit_rb": (90, True, None, "src/oracles/profit_oracle.rs:202-206"),
    "profit_q": (29, True, None, "src/oracles/profit_oracle.rs:220-224"),
    "cp_feas": (0, True, None, "tests/cutting_plane_tests.rs:134-135"),
    "cp_feas_no_soln": (2, False, None, "tests/cutting_plane_tests.rs:144-145"),
    "cp_optim": (None, True, None, "tests/cutting_plane_tests.rs:155"),
    "cp_optim_no_soln": (0, False, None, "tests/cutting_plane_tests.rs:164-165"),
    "cp_optim_max_iters": (2, False, None, "tests/cutting_plane_tests.rs:174-175"),
    "cp_feas_max_iters": (2, False, None, "tests/cutting_plane_tests.rs:184-185"),
    "cp_optim_q": (None, True, None, "tests/cutting_plane_tests.rs:281"),
    "cp_optim_q_no_soln": (0, False, None, "tests/cutting_plane_tests.rs:291-292"),
    "cp_optim_q_no_effect": (2, False, None, "tests/cutting_plane_tests.rs:301-302"),
    "bsearch": (30, None, 1, "tests/cutting_plane_tests.rs:315-316"),
    "bsearch_no_soln": (20, None, 0, "tests/cutting_plane_tests.rs:325-326"),
    "bsearch_adaptor": (None, None, 1, "tests/cutting_plane_tests.rs:369"),
    "example2_feasible": (1, True, None, "tests/example2_tests.rs:56-57"),
    "example2_infeasible": (0, False, None, "tests/example2_tests.rs:66-67"),
    "quad_n5": (None, True, None, "tests/integration_test.rs:118"),
    # benches/ellipsoid.rs: degenerate start: exit at iteration 0 with a NaN state
    "bench_degenerate_n10": (0, False, 1, "benches/ellipsoid.rs:34-39 + src/cutting_plane.rs:308"),
    "bench_degenerate_n50": (0, False, 1, "benches/ellipsoid.rs:34-39"),
    "bench_degenerate_n100": (0, False, 1, "benches/ellipsoid.rs:34-39"),
}

STABLE_CASES = [c for c in PINNED if c.endswith("_stable")]


def check_case(name, got):
    niter, has_x, flag, src = PINNED[name]
    if niter is not None:
        assert got["niter"] == niter, f"{name}: niter {got['niter']} != {niter} ({src})"
    if has_x is not None:
        assert got["has_x"] == has_x, f"{name}: has_x {got['has_x']} ({src})"
    if flag is not None:
        assert got["flag"] == flag, f"{name}: flag {got['flag']} ({src})"


def check_extra(results):
    """The inequality assertions that accompany the counts in the reference tests."""
    x = results["quasicvx_feasible"]["x"]               # src/quasicvx.rs:73-76
    assert 0.49 <= x[0] * x[0] <= 0.51 and 1.6 <= math.exp(x[1]) <= 1.7
    for c in ("profit", "profit_rb", "profit_q"):       # src/oracles/profit_oracle.rs:185,204,222
        assert results[c]["x"][0] <= math.log(30.5)
    x = results["quad_n5"]["x"]                          # tests/integration_test.rs:119-131
    rms = math.sqrt(sum((x[i] - (i + 1)) ** 2 for i in range(5)) / 5)
    assert rms < 3.0
