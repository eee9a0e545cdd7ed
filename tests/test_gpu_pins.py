"""GPU: the same end-to-end cases through the C++ host drivers with the MI355X engine as the search
space (C ABI); iteration counts must equal the reference's pins and solutions must match the
oracle-backed run."""
import numpy as np
import pytest

import pins
from cpp_build import build_runner, run_json_lines

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def results(gpu):
    return run_json_lines(build_runner("pins_runner.cpp", "hip"))


@pytest.fixture(scope="module")
def results_oracle():
    return run_json_lines(build_runner("pins_runner.cpp", "oracle"))


@pytest.mark.parametrize("case", sorted(pins.PINNED))
def test_pinned_case_hip_backend(results, case):
    pins.check_case(case, results[case])


def test_pinned_extra_assertions(results):
    pins.check_extra(results)


@pytest.fixture(scope="module")
def results_pipelined(gpu):
    return run_json_lines(build_runner("pins_runner.cpp", "hip"), "--pipelined")


@pytest.mark.parametrize("case", sorted(pins.PINNED))
def test_pinned_case_hip_pipelined_drivers(results_pipelined, case):
    """cutting_plane_optim_pipelined / _feas_pipelined (prime / cut / commit) reproduce the same pins."""
    pins.check_case(case, results_pipelined[case])


def test_pipelined_drivers_equal_plain_drivers(results, results_pipelined):
    for case, want in results.items():
        got = results_pipelined[case]
        assert (got["niter"], got["has_x"], got["flag"]) == (want["niter"], want["has_x"], want["flag"]), case
        assert got["x"] == want["x"] and got["gamma"] == want["gamma"], case   # bit-identical engine paths


@pytest.fixture(scope="module")
def results_deferred(gpu):
    return run_json_lines(build_runner("pins_runner.cpp", "hip"), "--pipelined", "--defer8")


@pytest.mark.parametrize("case", sorted(pins.PINNED))
def test_pinned_case_hip_deferred_shrink(results_deferred, case):
    """Deferred shrink (depth 8) + pipelined drivers still reproduce every pinned answer."""
    pins.check_case(case, results_deferred[case])


def test_solutions_match_oracle_backend(results, results_oracle):
    for case, want in results_oracle.items():
        got = results[case]
        assert got["niter"] == want["niter"], case
        assert got["has_x"] == want["has_x"] and got["flag"] == want["flag"], case
        if want["x"]:
            # quad_*: deep cuts with beta = f drive the ellipsoid to a numerically singular shape before
            # the 1e-10 exit, so last-bit differences in the dot-product order are amplified (iteration
            # counts still agree); the reference itself only asserts a loose error bound there
            # (tests/integration_test.rs:126-131, checked in pins.check_extra).
            if case.startswith("quad_"):
                continue
            np.testing.assert_allclose(got["x"], want["x"], rtol=1e-9, atol=1e-12, err_msg=case)
        if want["gamma"]:
            assert abs(got["gamma"] - want["gamma"]) <= 1e-9 * abs(want["gamma"]), case


def test_cpp_host_queue_replay(gpu):
    """EllHip::queue_upload / queue_run / queue_results (host/ellhip/ell_hip.hpp): a recorded cut sequence through the
    pipelined queue run (lookahead 16, groups on the matrix cores) against the same cuts taken one update at a time."""
    out = run_json_lines(build_runner("pins_runner.cpp", "hip"), "--queue-replay")["queue_replay"]
    assert out["ok"] is True and out["worst"] <= 1e-12, out
