"""GPU: BASELINE.json's full sizes (n = 16384; 2 GiB of Q), checked through size-independent properties
instead of the (too slow) CPU oracle:
  * closed form of the first update from Q0 = I (gt = g exactly): the whole matrix against numpy's
    elementwise (ratio*g_hi)*g_lo with the oracle's coefficient stage -- bit-exact up to omega's rounding;
  * symmetry preserved to the bit after further updates (the reference mirrors the lower triangle);
  * the GEMV of the next update against a numpy matvec of the downloaded matrix (tsq = kappa * g'Qg);
  * the pipelined schedule gives the same bits as the two-pass schedule.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 16384


@pytest.fixture(scope="module")
def cuts():
    from ellalgo_rs_amd import synth
    return synth.parallel_cuts(N, 6)


def test_first_update_closed_form_and_symmetry(gpu, orc, cuts):
    kinds, grads, b0, b1 = cuts
    e = gpu.Ell.new_with_scalar(1.0, np.zeros(N))
    g = grads[0]
    st = e._update(int(kinds[0]), (g, (b0[0], b1[0])))
    assert int(st) == 0
    omega = e.tsq()                       # kappa0 = 1  =>  tsq = omega = g.g (tree-summed on the device)
    assert abs(omega - float(g @ g)) <= 1e-13
    so, (rho, sigma, delta) = orc.Calc(N).dispatch(int(kinds[0]), b0[0], b1[0], omega)
    assert so == 0
    assert e.kappa == 1.0 * delta
    np.testing.assert_array_equal(e.xc(), 0.0 - (rho / omega) * g)
    ratio = sigma / omega
    q = e.mq
    rg = ratio * g
    # row by row to keep host memory bounded: Q1[r][c] = delta_rc - (ratio*g[max])*g[min]
    for r in range(0, N, 1024):
        rows = np.arange(r, r + 1024)[:, None]
        cols = np.arange(N)[None, :]
        upd = np.where(cols <= rows, rg[rows] * g[cols], rg[cols] * g[rows])
        want = (cols == rows).astype(np.float64) - upd
        np.testing.assert_array_equal(q[r:r + 1024], want)
    del q
    # a few more updates: symmetric to the bit, and the next GEMV agrees with numpy on the downloaded matrix
    for i in range(1, 4):
        assert int(e._update(int(kinds[i]), (grads[i], (b0[i], b1[i])))) == 0
    q = e.mq
    assert np.array_equal(q, q.T)
    kappa = e.kappa
    gn = grads[4]
    want_tsq = kappa * float(gn @ (q @ gn))
    assert int(e._update(int(kinds[4]), (gn, (b0[4], b1[4])))) == 0
    assert abs(e.tsq() - want_tsq) <= 1e-11 * abs(want_tsq)


def test_pipelined_equals_two_pass_at_full_size(gpu, cuts):
    kinds, grads, b0, b1 = cuts
    a = gpu.Ell.new_with_scalar(1.0, np.zeros(N))
    b = gpu.Ell.new_with_scalar(1.0, np.zeros(N))
    k = len(kinds)
    a.queue_upload(kinds, grads, b0, b1)
    b.queue_upload(kinds, grads, b0, b1)
    a.queue_run(0, k)
    b.queue_run(0, k, fused=True)
    sa, ta = a.queue_results()
    sb, tb = b.queue_results()
    assert np.array_equal(sa, sb) and np.all(sa == 0) and np.array_equal(ta, tb)
    assert a.kappa == b.kappa and np.array_equal(a.xc(), b.xc())
    qa = a.mq
    assert np.array_equal(qa, b.mq)
    assert np.array_equal(qa, qa.T)


def test_deferred_symv_equals_immediate_path_at_full_size(gpu, cuts):
    """Default fast configuration (depth 8, lower-triangle GEMV, pipelined) against the reference data flow
    (depth 1, two-pass) on the same cuts: same statuses, state equal to rounding."""
    from ellalgo_rs_amd import synth
    k = 20
    kinds, grads, b0, b1 = synth.parallel_cuts(N, k)
    a = gpu.Ell.new_with_scalar(1.0, np.zeros(N))
    b = gpu.Ell.new_with_scalar(1.0, np.zeros(N))
    b.defer_depth = 8
    a.queue_upload(kinds, grads, b0, b1)
    b.queue_upload(kinds, grads, b0, b1)
    a.queue_run(0, k)
    b.queue_run(0, k, fused=True)
    sa, ta = a.queue_results()
    sb, tb = b.queue_results()
    assert np.array_equal(sa, sb) and np.all(sa == 0)
    assert np.max(np.abs(ta - tb) / np.abs(ta)) <= 1e-12
    assert abs(a.kappa - b.kappa) <= 1e-12 * abs(a.kappa)
    assert np.max(np.abs(a.xc() - b.xc())) <= 1e-12 * np.max(np.abs(a.xc()))
    qa, qb = a.mq, b.mq
    assert np.array_equal(qb, qb.T)
    assert np.max(np.abs(qa - qb)) <= 1e-12 * np.max(np.abs(qa))


def test_ellstable_first_update_closed_form(gpu, orc):
    """EllStable from the identity factor: w = g, z = g, every parked product U[j][i]*w[j] is 0, so the
    off-diagonal part of the buffer stays exactly zero and only the diagonal is rescaled (t_{j-1}/t_j)."""
    n = 4096
    from ellalgo_rs_amd import synth
    kinds, grads, b0, _ = synth.deep_cuts(n, 1)
    e = gpu.EllStable.new_with_scalar(1.0, np.zeros(n))
    o = orc.OracleEllStable.new_with_scalar(1.0, np.zeros(n))
    assert int(e.update_bias_cut((grads[0], float(b0[0])))) == o.update(0, grads[0], b0[0]) == 0
    m = e.mq
    assert np.count_nonzero(m - np.diag(np.diag(m))) == 0        # off-diagonal stays exactly zero
    np.testing.assert_allclose(np.diag(m), np.diag(o.mq), rtol=1e-12)
    np.testing.assert_allclose(e.xc(), o.xc, rtol=1e-12, atol=1e-300)
    assert abs(e.kappa - o.kappa) <= 1e-13 * abs(o.kappa)
