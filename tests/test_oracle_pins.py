"""CPU: pins the oracle (oracle/ell_oracle.c) with the reference's own one-step and coefficient known
answers, so that it can serve as the checker for the HIP path."""
import numpy as np
import pytest

from test_gpu_parity_ell import CALC_CASES


def approx(a, b, eps=1e-6):  # approx_eq::assert_approx_eq! default
    return abs(a - b) <= eps * max(abs(a), abs(b)) or abs(a - b) < eps


# ------------------------------------------------------------------ src/ell_calc.rs:942-1186
def test_calc_core_construct(orc):  # :944-950
    c = orc.Calc(4).c
    assert c.n_f == 4.0 and c.half_n == 2.0 and c.n_plus_1 == 5.0
    assert approx(c.cst1, 16.0 / 15.0) and approx(c.cst2, 0.4)


def test_calc_core_fast_forms(orc):  # :953-970
    calc = orc.Calc(4)
    assert all(approx(g, w) for g, w in zip(calc.core_parallel_bias_cut_fast(1.0, 2.0, 4.0, 2.0, 12.0), (1.2, 0.8, 0.8)))
    assert all(approx(g, w) for g, w in zip(calc.core_bias_cut_fast(1.0, 2.0, 6.0), (1.2, 0.8, 0.8)))


@pytest.mark.parametrize("kind,beta,tsq,status,want", CALC_CASES)
def test_calc_known_answers(orc, kind, beta, tsq, status, want):
    b0, b1 = beta if isinstance(beta, tuple) else (beta, None)
    st, got = orc.Calc(4).dispatch(kind, b0, b1, tsq)
    assert st == status
    if want is not None:
        for g, w in zip(got, want):
            if w is not None:
                assert approx(g, w)


def test_calc_exact_values(orc):  # src/ell_calc_additional_tests.rs: sigma == 0.4, delta == 16/15 exactly
    st, (rho, sigma, delta) = orc.Calc(4).calc_central_cut(0.0)
    assert st == 0 and rho == 0.0 and sigma == 0.4 and delta == 16.0 / 15.0


def test_calc_use_parallel_cut_false_falls_back(orc):  # src/ell_calc.rs:761,797,840
    n = 4
    e = orc.OracleEll.new_with_scalar(0.01, np.zeros(n))
    f = orc.OracleEll.new_with_scalar(0.01, np.zeros(n))
    e.set_use_parallel_cut(False)
    g = 0.5 * np.ones(n)
    assert e.update_bias_cut(g, 0.01, 0.04) == 0 and f.update_bias_cut(g, 0.01) == 0
    assert np.array_equal(e.mq, f.mq) and e.kappa == f.kappa


# ------------------------------------------------------------------ src/ell.rs:236-364
@pytest.fixture
def ell4(orc):
    return orc.OracleEll.new_with_scalar(0.01, np.zeros(4))


def test_ell_construct(orc, ell4):
    assert approx(ell4.kappa, 0.01) and np.array_equal(ell4.mq, np.eye(4)) and ell4.tsq == 0.0


def test_ell_update_central_cut(ell4):  # :247-256, xc and mq bit-exact
    assert ell4.update_central_cut(0.5 * np.ones(4)) == 0
    assert np.array_equal(ell4.xc, -0.01 * np.ones(4))
    assert np.array_equal(ell4.mq, np.eye(4) - 0.1 * np.ones((4, 4)))
    assert approx(ell4.kappa, 0.16 / 15.0) and approx(ell4.tsq, 0.01)


def test_ell_update_bias_cut(ell4):  # :259-268
    assert ell4.update_bias_cut(0.5 * np.ones(4), 0.05) == 0
    assert approx(ell4.xc[0], -0.03) and approx(ell4.mq[0, 0], 0.8) and approx(ell4.kappa, 0.008)


def test_ell_update_parallel_central_cut(ell4):  # :271-280, bit-exact
    assert ell4.update_central_cut(0.5 * np.ones(4), 0.0, 0.05) == 0
    assert np.array_equal(ell4.xc, -0.01 * np.ones(4))
    assert np.array_equal(ell4.mq, np.eye(4) - 0.2 * np.ones((4, 4)))
    assert approx(ell4.kappa, 0.012)


def test_ell_update_parallel(ell4):  # :283-292
    assert ell4.update_bias_cut(0.5 * np.ones(4), 0.01, 0.04) == 0
    assert approx(ell4.xc[0], -0.0116) and approx(ell4.mq[0, 0], 1.0 - 0.232) and approx(ell4.kappa, 0.01232)


def test_ell_update_parallel_no_effect(ell4):  # :295-303, bit-exact
    assert ell4.update_bias_cut(0.5 * np.ones(4), -0.04, 0.0625) == 0
    assert np.array_equal(ell4.xc, np.zeros(4)) and np.array_equal(ell4.mq, np.eye(4)) and approx(ell4.kappa, 0.01)


def test_ell_update_q_no_effect(ell4):  # :306-314
    assert ell4.update_q(0.5 * np.ones(4), -0.04, 0.0625) == 2
    assert np.array_equal(ell4.xc, np.zeros(4)) and np.array_equal(ell4.mq, np.eye(4))


def test_ell_update_q(ell4):  # :317-326
    assert ell4.update_q(0.5 * np.ones(4), 0.01, 0.04) == 0
    assert approx(ell4.xc[0], -0.0116) and approx(ell4.mq[0, 0], 1.0 - 0.232) and approx(ell4.kappa, 0.01232)


def test_ell_no_defer_trick(ell4):  # :342-354
    ell4.set_no_defer_trick(True)
    ell4.update_central_cut(0.5 * np.ones(4))
    assert approx(ell4.kappa, 1.0)
    assert np.allclose(ell4.mq, (np.eye(4) - 0.1 * np.ones((4, 4))) * (0.16 / 15.0), rtol=1e-6)


# ------------------------------------------------------------------ src/ell_stable.rs:217-307
@pytest.fixture
def stable4(orc):
    return orc.OracleEllStable.new_with_scalar(0.01, np.zeros(4))


def test_stable_central(stable4):  # :226-234
    assert stable4.update_central_cut(0.5 * np.ones(4)) == 0
    assert np.array_equal(stable4.xc, -0.01 * np.ones(4))
    assert approx(stable4.kappa, 0.16 / 15.0) and approx(stable4.tsq, 0.01)


def test_stable_bias(stable4):  # :237-245
    assert stable4.update_bias_cut(0.5 * np.ones(4), 0.05) == 0
    assert approx(stable4.xc[0], -0.03) and approx(stable4.kappa, 0.008)


def test_stable_parallel_central(stable4):  # :248-256
    assert stable4.update_central_cut(0.5 * np.ones(4), 0.0, 0.05) == 0
    assert np.array_equal(stable4.xc, -0.01 * np.ones(4)) and approx(stable4.kappa, 0.012)


def test_stable_parallel(stable4):  # :259-267
    assert stable4.update_bias_cut(0.5 * np.ones(4), 0.01, 0.04) == 0
    assert approx(stable4.xc[0], -0.0116) and approx(stable4.kappa, 0.01232)


def test_stable_parallel_no_effect(stable4):  # :270-277
    assert stable4.update_bias_cut(0.5 * np.ones(4), -0.04, 0.0625) == 0
    assert np.array_equal(stable4.xc, np.zeros(4)) and approx(stable4.kappa, 0.01)


def test_stable_q(stable4):  # :280-298
    assert stable4.update_q(0.5 * np.ones(4), -0.04, 0.0625) == 2
    assert stable4.update_q(0.5 * np.ones(4), 0.01, 0.04) == 0
    assert approx(stable4.xc[0], -0.0116) and approx(stable4.kappa, 0.01232)


# ------------------------------------------------------------------ internal consistency of the oracle
def test_rowwise_form_is_bit_identical_for_symmetric_q(orc):
    """(ratio*gt[max])*gt[min] evaluated row-wise == the reference's lower-triangle + mirror loop."""
    n = 37
    rng = np.random.default_rng(0)
    a = orc.OracleEll.new_with_scalar(1.0, np.zeros(n))
    b = orc.OracleEll.new_with_scalar(1.0, np.zeros(n))
    for i in range(30):
        g = rng.standard_normal(n)
        assert a.update(0, g, 0.01) == b.update_rowwise(0, g, 0.01) == 0
        assert np.array_equal(a.mq, b.mq) and np.array_equal(a.xc, b.xc) and a.kappa == b.kappa


def test_stable_corrected_variant_tracks_ell(orc):
    """with both defects corrected EllStable is algebraically Ell; as written it is not."""
    n = 6
    rng = np.random.default_rng(4)
    e = orc.OracleEll.new_with_scalar(1.0, np.zeros(n))
    s = orc.OracleEllStable.new_with_scalar(1.0, np.zeros(n))
    c = orc.OracleEllStable.new_with_scalar(1.0, np.zeros(n))
    c.set_corrected(True)
    for i in range(30):
        g = rng.standard_normal(n)
        e.update(0, g, 0.01), s.update(0, g, 0.01), c.update(0, g, 0.01)
    assert abs(c.tsq - e.tsq) <= 1e-12 * abs(e.tsq)
    assert np.allclose(c.xc, e.xc, rtol=1e-10, atol=1e-13)
    assert abs(s.tsq - e.tsq) > 1e-6 * abs(e.tsq)   # the reference's variant really differs


def test_row_block_gemv_equals_full(orc):
    n = 48
    rng = np.random.default_rng(2)
    q = rng.standard_normal((n, n))
    g = rng.standard_normal(n)
    full = np.zeros(n)
    orc.rows_gemv(n, 0, n, q, g, full)
    parts = np.zeros(n)
    orc.rows_gemv(n, 0, 16, q[:16], g, parts)
    orc.rows_gemv(n, 16, 32, q[16:], g, parts)
    assert np.array_equal(full, parts)


def test_rowwise_openmp_variant_is_bit_identical_to_the_serial_row_wise_form():
    """bench.py's "all cores" CPU line (not the reference's loop) must still be the reference's arithmetic."""
    import numpy as np
    from oracle import oracle as O
    n = 257
    rng = np.random.default_rng(3)
    a, b = O.OracleEll.new_with_scalar(2.0, np.linspace(-1, 1, n)), O.OracleEll.new_with_scalar(2.0, np.linspace(-1, 1, n))
    for i in range(12):
        g = rng.standard_normal(n)
        beta = 0.02 * (i % 3)
        assert a.update_rowwise(0, g, beta) == b.update_rowwise_mt(0, g, beta)
    assert np.array_equal(a.mq, b.mq) and np.array_equal(a.xc, b.xc) and a.kappa == b.kappa and a.tsq == b.tsq


def test_three_oracle_loops_agree_bit_for_bit_at_n2048():
    """The full-size GPU tests (tests/test_gpu_fullsize.py, n = 16384 / 32768) check against `update_rowwise_mt`
    -- the row-parallel OpenMP form of the same arithmetic -- because the reference's loop order (column-strided mirror
    stores, src/ell.rs:117-128) takes seconds per update there.  This pins that checker to the reference loop at a size
    where all three still run quickly: `update` (reference loop order), `update_rowwise` and `update_rowwise_mt` must
    leave bit-identical Q, xc, kappa, tsq and statuses over a mixed sequence (all six EllCalc entry points, one
    failing cut)."""
    import numpy as np
    from oracle import oracle as O
    from util import mixed_cut
    n = 2048
    rng = np.random.default_rng(2048)
    xc0 = np.linspace(-1.0, 1.0, n)
    a, b, c = (O.OracleEll.new_with_scalar(2.0, xc0) for _ in range(3))
    for i in range(8):
        g = rng.standard_normal(n)
        g /= np.linalg.norm(g)
        tau = float(np.sqrt(a.kappa * (g @ (a.mq @ g))))
        kind, b0, b1 = mixed_cut(i, g, tau, rng)
        sa, sb, sc = a.update(kind, g, b0, b1), b.update_rowwise(kind, g, b0, b1), c.update_rowwise_mt(kind, g, b0, b1)
        assert sa == sb == sc == (1 if i == 7 else 0)
        assert a.tsq == b.tsq == c.tsq and a.kappa == b.kappa == c.kappa
    assert np.array_equal(a.mq, b.mq) and np.array_equal(a.mq, c.mq)
    assert np.array_equal(a.xc, b.xc) and np.array_equal(a.xc, c.xc)
