"""CPU: the oracle + the C++ host drivers reproduce every pinned end-to-end answer of the reference."""
import pytest

import pins
from cpp_build import build_runner, run_json_lines


@pytest.fixture(scope="module")
def results():
    return run_json_lines(build_runner("pins_runner.cpp", "oracle"))


@pytest.mark.parametrize("case", sorted(pins.PINNED))
def test_pinned_case_oracle_backend(results, case):
    pins.check_case(case, results[case])


def test_pinned_extra_assertions(results):
    pins.check_extra(results)


def test_config1_n16_plumbing(results):
    """BASELINE config 1: cutting_plane_optim + the quadratic oracle at n = 16 on the CPU path."""
    r = results["quad_n16"]
    assert r["has_x"] and 0 < r["niter"] < 2000
