"""GPU parity tests for EllStable (src/ell_stable.rs): HIP engine vs the bug-compatible CPU oracle and
the reference's one-step known answers (src/ell_stable.rs:217-307)."""
import numpy as np
import pytest

from util import assert_state_close, run_mixed

pytestmark = pytest.mark.gpu


def _approx(a, b, eps=1e-6):
    return abs(a - b) <= eps * max(abs(a), abs(b), 1e-300) or abs(a - b) < eps


def test_construct(gpu):  # :218-223
    e = gpu.EllStable.new_with_scalar(0.01, np.zeros(4))
    assert _approx(e.kappa, 0.01) and np.array_equal(e.xc(), np.zeros(4)) and e.tsq() == 0.0


def test_new(gpu):  # :301-307
    e = gpu.EllStable.new(np.array([1.0, 1.0]), np.zeros(2))
    assert e.kappa == 1.0 and np.array_equal(e.mq, np.eye(2))


def test_update_central_cut(gpu):  # :226-234 (xc is assert_eq!)
    e = gpu.EllStable.new_with_scalar(0.01, np.zeros(4))
    assert e.update_central_cut((0.5 * np.ones(4), gpu.SingleCut(0.0))) == 0
    assert np.array_equal(e.xc(), -0.01 * np.ones(4))
    assert _approx(e.kappa, 0.16 / 15.0) and _approx(e.tsq(), 0.01)


def test_update_bias_cut(gpu):  # :237-245
    e = gpu.EllStable.new_with_scalar(0.01, np.zeros(4))
    assert e.update_bias_cut((0.5 * np.ones(4), gpu.SingleCut(0.05))) == 0
    assert _approx(e.xc()[0], -0.03) and _approx(e.kappa, 0.008) and _approx(e.tsq(), 0.01)


def test_update_parallel_central_cut(gpu):  # :248-256
    e = gpu.EllStable.new_with_scalar(0.01, np.zeros(4))
    assert e.update_central_cut((0.5 * np.ones(4), gpu.ParallelCut(0.0, 0.05))) == 0
    assert np.array_equal(e.xc(), -0.01 * np.ones(4)) and _approx(e.kappa, 0.012)


def test_update_parallel(gpu):  # :259-267
    e = gpu.EllStable.new_with_scalar(0.01, np.zeros(4))
    assert e.update_bias_cut((0.5 * np.ones(4), gpu.ParallelCut(0.01, 0.04))) == 0
    assert _approx(e.xc()[0], -0.0116) and _approx(e.kappa, 0.01232)


def test_update_parallel_no_effect(gpu):  # :270-277
    e = gpu.EllStable.new_with_scalar(0.01, np.zeros(4))
    assert e.update_bias_cut((0.5 * np.ones(4), gpu.ParallelCut(-0.04, 0.0625))) == 0
    assert np.array_equal(e.xc(), np.zeros(4)) and _approx(e.kappa, 0.01)


def test_update_q(gpu):  # :280-298
    e = gpu.EllStable.new_with_scalar(0.01, np.zeros(4))
    assert e.update_q((0.5 * np.ones(4), gpu.ParallelCut(-0.04, 0.0625))) == gpu.CutStatus.NoEffect
    assert np.array_equal(e.xc(), np.zeros(4)) and _approx(e.kappa, 0.01)
    assert e.update_q((0.5 * np.ones(4), gpu.ParallelCut(0.01, 0.04))) == 0
    assert _approx(e.xc()[0], -0.0116) and _approx(e.kappa, 0.01232)


@pytest.mark.parametrize("n", [2, 3, 5, 16, 63, 64, 65, 127, 128, 130, 200, 257, 512, 1000])
def test_mixed_sequence_matches_oracle(gpu, orc, n):
    """Whole buffer (diag, factor AND scratch triangle), xc, kappa, tsq after a mixed cut sequence."""
    xc0 = np.linspace(-1.0, 1.0, n)
    g = gpu.EllStable.new_with_scalar(2.0, xc0)
    o = orc.OracleEllStable.new_with_scalar(2.0, xc0)
    nsucc = run_mixed(g, o, 24, seed=300 + n, check_every=6)
    assert nsucc >= 12
    assert_state_close(g, o, what=f"n={n} final")


@pytest.mark.parametrize("n", [1024, 2048])
def test_deep_cuts_large_matches_oracle(gpu, orc, n):
    from ellalgo_rs_amd import synth
    kinds, grads, b0, _ = synth.deep_cuts(n, 5)
    g = gpu.EllStable.new_with_scalar(1.0, np.zeros(n))
    o = orc.OracleEllStable.new_with_scalar(1.0, np.zeros(n))
    for i in range(5):
        assert int(g.update_bias_cut((grads[i], float(b0[i])))) == o.update(0, grads[i], b0[i]) == 0
    assert_state_close(g, o, what=f"n={n}")


def test_failed_cut_rewrites_only_the_scratch_triangle(gpu, orc):  # src/ell_stable.rs:66,88-90
    n = 70
    rng = np.random.default_rng(5)
    g = gpu.EllStable.new_with_scalar(1.0, np.zeros(n))
    o = orc.OracleEllStable.new_with_scalar(1.0, np.zeros(n))
    for i in range(3):
        gr = rng.standard_normal(n)
        assert int(g.update_bias_cut((gr, 0.01))) == o.update(0, gr, 0.01) == 0
    gr = rng.standard_normal(n)
    k0, x0 = g.kappa, g.xc()
    upper0 = np.triu(g.mq)
    assert int(g.update_bias_cut((gr, 1e6))) == o.update(0, gr, 1e6) == 1
    assert g.kappa == k0 and np.array_equal(g.xc(), x0)
    assert np.array_equal(np.triu(g.mq), upper0)          # diag + factor untouched
    assert_state_close(g, o, what="after failed cut")      # scratch triangle matches the oracle's


def test_clone_and_queue(gpu, orc):
    from ellalgo_rs_amd import synth
    n, k = 192, 8
    kinds, grads, b0, b1 = synth.deep_cuts(n, k)
    a = gpu.EllStable.new_with_scalar(1.0, np.zeros(n))
    a.queue_upload(kinds, grads, b0, b1)
    a.queue_run(0, k)
    st, ts = a.queue_results()
    assert list(st) == [0] * k
    b = gpu.EllStable.new_with_scalar(1.0, np.zeros(n))
    for i in range(4):
        b.update_bias_cut((grads[i], b0[i]))
    c = b.clone()
    for i in range(4, k):
        b.update_bias_cut((grads[i], b0[i]))
        c.update_bias_cut((grads[i], b0[i]))
    assert np.array_equal(a.mq, b.mq) and np.array_equal(b.mq, c.mq)
    assert np.array_equal(a.xc(), b.xc()) and a.kappa == b.kappa == c.kappa
