"""GPU: the reference's loose end-to-end sanity tests (tests/integration_test.rs, tests/regression_tests.rs,
tests/quasicvx2_tests.rs; n = 2 ... 8) with the HIP engines as the search space: the reference's own assertions,
and -- on the batched engine, which is bit-identical to the CPU arithmetic -- the very iteration counts, gammas
and solutions of the same loops on the CPU oracle."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def sum_sq_oracle(target=None):
    """f = |x - t|^2 with the cut value f itself (integration_test.rs:12-22, regression_tests.rs:181-195)"""
    def assess(x, gamma):
        d = x if target is None else x - target
        f = 0.0
        for v in d.tolist():
            f += v * v
        g = 2.0 * d
        if f < gamma:
            return (g, f), True, f
        return (g, f), False, gamma
    return assess


class Quasicvx2:
    """tests/quasicvx2_tests.rs:15-72 (no round robin reset between calls: idx persists)"""

    def __init__(self):
        self.idx = -1

    def __call__(self, xc, gamma):
        x, y = float(xc[0]), float(xc[1])
        for _ in range(3):
            self.idx += 1
            if self.idx == 3:
                self.idx = 0
            if self.idx == 0:
                tmp = math.exp(x)
                grad, fj = np.array([tmp, -1.0]), tmp - y
            elif self.idx == 1:
                grad, fj = np.array([0.0, -1.0]), -y
            else:
                grad, fj = np.array([-1.0, 0.0]), -x
            if fj > 0.0:
                return (grad, fj), False, gamma
        tmp2 = math.sqrt(x)
        fj = -tmp2 + gamma * y
        if fj > 0.0:
            return (np.array([-0.5 / tmp2, gamma]), fj), False, gamma
        gamma = tmp2 / y
        return (np.array([-0.5 / tmp2, gamma]), 0.0), True, gamma


def solve(gpu, orc, space, kappa, x0, make_oracle, max_iters, tol, gamma0=math.inf):
    """the same cutting_plane_optim loop on the chosen engine and on the CPU oracle"""
    x0 = np.asarray(x0, dtype=np.float64)
    o = orc.OracleEll.new_with_scalar(kappa, x0)
    want = run_loop_g(lambda k, g, b: o.update(k, g, b), lambda: np.array(o.xc), lambda: o.tsq, make_oracle(), max_iters,
                      tol, gamma0)
    if space == "batch":
        e = gpu.EllBatch.new_with_scalar(kappa, x0[None, :])
        last = {}

        def upd(k, g, b):
            st, ts = e.update(np.array([k], dtype=np.int32), g[None, :], np.array([b]))
            last["tsq"] = float(ts[0, 0])
            return int(st[0, 0])
        got = run_loop_g(upd, lambda: e.xc()[0], lambda: last["tsq"], make_oracle(), max_iters, tol, gamma0)
        assert got[1] == want[1] and got[2] == want[2]
        assert (got[0] is None) == (want[0] is None) and (got[0] is None or np.array_equal(got[0], want[0]))
    else:
        e = gpu.Ell.new_with_scalar(kappa, x0)
        got = run_loop_g(lambda k, g, b: int(e._update(k, (g, b))), e.xc, e.tsq, make_oracle(), max_iters, tol, gamma0)
        assert (got[0] is None) == (want[0] is None)
    return got


def run_loop_g(update, xc, tsq, ask, max_iters, tol, gamma0):
    gamma, x_best = gamma0, None
    for niter in range(max_iters):
        x = xc()
        (g, beta), shrunk, gamma = ask(x, gamma)
        if shrunk:
            x_best = x
        st = update(1 if shrunk else 0, g, beta)
        if st != 0 or tsq() < tol:
            return x_best, niter, gamma
    return x_best, max_iters, gamma


@pytest.mark.parametrize("space", ["ell", "batch"])
def test_integration_simple_quadratic_and_known_optimum(gpu, orc, space):
    xb, _, gamma = solve(gpu, orc, space, 10.0, [5.0, 5.0], sum_sq_oracle, 1000, 1e-10)   # integration_test.rs:6-37
    assert xb is not None and abs(xb[0]) < 0.5 and abs(xb[1]) < 0.5 and gamma < 1.0
    target = np.array([2.0, -1.5])                                                      # :40-82
    xb, _, gamma = solve(gpu, orc, space, 20.0, [10.0, 10.0], lambda: sum_sq_oracle(target), 1000, 1e-10)
    assert xb is not None and math.hypot(xb[0] - target[0], xb[1] - target[1]) < 15.0
    for x0 in ([1.0, 1.0], [-1.0, -1.0], [5.0, -5.0], [10.0, 0.0]):                     # :172-220
        xb, _, gamma = solve(gpu, orc, space, 10.0, x0, sum_sq_oracle, 1000, 1e-10)
        assert xb is not None and gamma < 100.0


@pytest.mark.parametrize("space", ["ell", "batch"])
def test_regression_iterations_and_dimensional_scaling(gpu, orc, space):
    xb, niter, gamma = solve(gpu, orc, space, 10.0, [3.0, 3.0], sum_sq_oracle, 1000, 1e-10)  # regression_tests.rs:6-41
    assert niter < 1000 and gamma < 10.0
    for ndim in (2, 4, 8):                                                                  # :172-213
        xb, niter, gamma = solve(gpu, orc, space, 10.0, [3.0] * ndim, sum_sq_oracle, 3000, 1e-10)
        assert xb is not None and niter < 3000


@pytest.mark.parametrize("space", ["ell", "batch"])
def test_quasicvx2_cases(gpu, orc, space):
    opts = (2000, 1e-20)  # Options::default()
    xb, _, _ = solve(gpu, orc, space, 10.0, [1.0, 1.0], Quasicvx2, *opts, gamma0=0.0)       # quasicvx2_tests.rs:75-82
    assert xb is not None
    xb, _, _ = solve(gpu, orc, space, 10.0, [100.0, 100.0], Quasicvx2, *opts, gamma0=0.0)   # :85-92
    assert xb is None
    xb, _, _ = solve(gpu, orc, space, 10.0, [1.0, 1.0], Quasicvx2, *opts, gamma0=100.0)     # :95-101
    assert xb is None
