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

from util import set_default

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


@pytest.mark.parametrize("lookahead", [3, 16, 32])
def test_pipelined_equals_two_pass_at_full_size(gpu, cuts, lookahead):
    """The pipelined queue run forms the products of up to `lookahead` queued cuts in one pass over Q: up to 3 on the
    vector ALU with k_symv's arithmetic (bit-identical to the two-pass run), the default 16 on the matrix cores with the
    group stage (own association: 1e-12)."""
    kinds, grads, b0, b1 = cuts
    a = gpu.Ell.new_with_scalar(1.0, np.zeros(N))
    b = gpu.Ell.new_with_scalar(1.0, np.zeros(N))
    b.set_option(gpu.capi.OPT_LOOKAHEAD, lookahead)
    k = len(kinds)
    a.queue_upload(kinds, grads, b0, b1)
    b.queue_upload(kinds, grads, b0, b1)
    a.queue_run(0, k)
    b.queue_run(0, k, fused=True)
    sa, ta = a.queue_results()
    sb, tb = b.queue_results()
    assert np.array_equal(sa, sb) and np.all(sa == 0)
    if lookahead <= 3:
        assert np.array_equal(ta, tb) and a.kappa == b.kappa and np.array_equal(a.xc(), b.xc())
    else:
        assert np.max(np.abs(ta - tb) / ta) <= 1e-12 and abs(a.kappa - b.kappa) <= 1e-12 * a.kappa
        assert np.max(np.abs(a.xc() - b.xc())) <= 1e-12 * np.max(np.abs(a.xc()))
    qa, qb = a.mq, b.mq
    if lookahead <= 3:
        assert np.array_equal(qa, qb)
    else:
        for r in range(0, N, 2048):   # (in blocks of rows: the difference of two 2 GiB matrices is a third)
            assert np.max(np.abs(qa[r:r + 2048] - qb[r:r + 2048])) <= 1e-12
    assert np.array_equal(qa, qa.T) and np.array_equal(qb, qb.T)


def test_deferred_symv_equals_immediate_path_at_full_size(gpu, cuts):
    """Default fast configuration (depth 8, lower-triangle GEMV, pipelined) against the reference data flow
    (depth 1, two-pass) on the same cuts: same statuses, state equal to rounding."""
    from ellalgo_rs_amd import synth
    k = 20
    kinds, grads, b0, b1 = synth.parallel_cuts(N, k)
    a = gpu.Ell.new_with_scalar(1.0, np.zeros(N))
    b = gpu.Ell.new_with_scalar(1.0, np.zeros(N))
    a.defer_depth = 1   # (a new handle of this size starts at depth 16)
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


def _close_in_blocks(qg, qo, tol, what):
    """max|qg - qo| <= tol * max|qo| without a third full-size temporary (row blocks)."""
    scale = 0.0
    err = 0.0
    for r in range(0, qo.shape[0], 1024):
        a, b = qg[r:r + 1024], qo[r:r + 1024]
        scale = max(scale, float(np.max(np.abs(b))))
        err = max(err, float(np.max(np.abs(a - b))))
    assert err <= tol * scale, f"{what}: Q abs err {err} vs scale {scale}"


def test_ell_default_schedule_matches_oracle_at_full_size(gpu, orc):
    """The timed configuration itself (n = 16384, parallel cuts, a depth-24 handle's pipelined queue run: the products of
    up to 32 queued cuts per pass over the lower triangle on the matrix cores, the group stage, up to 48 recorded updates applied
    as one rank-48 update) against the CPU oracle on the same 60 cuts -- groups of 32, 16, an apply pass at cut 48,
    a group of 12 still recorded when the state is read.  Whole state to the north-star tolerance.
    (The checker is the oracle's row-parallel loop `update_rowwise_mt`, used here for speed only: the reference's loop
    order takes seconds per update at this size; tests/test_oracle_pins.py ties it bit for bit to the reference loop
    at n = 37, 257 and 2048.)"""
    from ellalgo_rs_amd import synth
    from util import TOL
    k = 60
    kinds, grads, b0, b1 = synth.parallel_cuts(N, k)
    e = gpu.Ell.new_with_scalar(1.0, np.zeros(N))
    assert e.defer_depth == 24          # what a new handle of this size starts with
    assert e.get_option(gpu.capi.OPT_LOOKAHEAD) == 32 and e.get_option(gpu.capi.OPT_QUEUE_DEPTH) == 48
    e.queue_upload(kinds, grads, b0, b1)
    e.queue_run(0, k, fused=True)
    st, ts = e.queue_results()
    o = orc.OracleEll.new_with_scalar(1.0, np.zeros(N))
    want_ts = np.empty(k)
    for i in range(k):
        assert o.update_rowwise_mt(int(kinds[i]), grads[i], b0[i], None if np.isnan(b1[i]) else b1[i]) == 0
        want_ts[i] = o.tsq
    assert np.all(st == 0)
    assert np.max(np.abs(ts - want_ts) / np.abs(want_ts)) <= TOL
    assert abs(e.kappa - o.kappa) <= TOL * abs(o.kappa)
    xo = np.array(o.xc)
    assert np.max(np.abs(e.xc() - xo)) <= TOL * np.max(np.abs(xo))
    qg = e.mq                      # applies the twelve recorded updates and mirrors the lower triangle
    _close_in_blocks(qg, o.mq, TOL, "n=16384 default queue run")
    assert np.array_equal(qg[:2048, :2048], qg[:2048, :2048].T)


@pytest.mark.parametrize("solve", [3])   # (2, the eager helped kernels: bit for bit against the plain ones up to n = 8200 in
def test_ellstable_matches_oracle_at_full_size(gpu, orc, solve):   # test_gpu_ellstable_factor.py; at this size in round 3)
    """BASELINE config 5 (n = 16384 EllStable, deep cuts) from the NON-trivial factor bench.py uses
    (synth.stable_factor: random unit-upper-triangular factor, random positive diagonal, junk in the scratch
    triangle -- from the identity the off-diagonal part of the buffer stays exactly zero and the comparison would be
    one of zeros): the persistent flag-chained solves over all 128 column strips against the serial CPU oracle --
    diagonal, factor AND scratch triangle (each on its own scale), xc, kappa, tsq."""
    from ellalgo_rs_amd import synth
    from util import TOL
    k = 3
    kinds, grads, b0, _ = synth.deep_cuts(N, k)
    f = synth.stable_factor(N)
    set_default("STABLE_SOLVE", solve)   # 3: the mirrored layout (three cuts inside it, the buffer rebuilt for the comparison)
    e = gpu.EllStable.new_with_matrix(1.0, f, np.zeros(N))
    o = orc.OracleEllStable.new_with_matrix(1.0, f, np.zeros(N))
    del f
    for i in range(k):
        assert int(e.update_bias_cut((grads[i], float(b0[i])))) == o.update(0, grads[i], b0[i]) == 0
        assert abs(e.tsq() - o.tsq) <= TOL * abs(o.tsq)
    assert abs(e.kappa - o.kappa) <= TOL * abs(o.kappa)
    xo = np.array(o.xc)
    assert np.max(np.abs(e.xc() - xo)) <= TOL * np.max(np.abs(xo))
    qg, qo = e.mq, o.mq
    _close_in_blocks(qg, qo, TOL, "n=16384 EllStable")
    # the triangles separately (the scratch products are ~1e-3 of the diagonal's scale), on row bands
    for r in (0, 4096 - 64, 8192 - 64, N - 1024):
        a, b = qg[r:r + 1024], qo[r:r + 1024]
        cols = np.arange(N)[None, :]
        rows = np.arange(r, r + a.shape[0])[:, None]
        for name, mask in (("factor", cols > rows), ("scratch", cols < rows)):
            if not mask.any():
                continue
            sc = float(np.max(np.abs(b[mask])))
            assert sc > 0.0, f"{name} rows {r}: the oracle's triangle is all zero"
            assert float(np.max(np.abs(a[mask] - b[mask]))) <= TOL * sc, f"{name} rows {r}"


def test_ell_n32768_matches_oracle(gpu, orc):
    """BASELINE config 4's size on ONE GPU (Q = 8 GiB): two deep cuts through the depth-16 schedule against the
    oracle; the 8 GiB matrices are compared on a band of rows (first, middle, last 512) to bound host memory."""
    from ellalgo_rs_amd import synth
    from util import TOL
    n, k = 32768, 2
    kinds, grads, b0, _ = synth.deep_cuts(n, k)
    e = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    e.defer_depth = 16
    o = orc.OracleEll.new_with_scalar(1.0, np.zeros(n))
    for i in range(k):
        assert int(e.update_bias_cut((grads[i], float(b0[i])))) == 0
        assert o.update_rowwise_mt(0, grads[i], b0[i], None) == 0
        assert abs(e.tsq() - o.tsq) <= TOL * abs(o.tsq)
    assert abs(e.kappa - o.kappa) <= TOL * abs(o.kappa)
    xo = np.array(o.xc)
    assert np.max(np.abs(e.xc() - xo)) <= TOL * np.max(np.abs(xo))
    qg, qo = e.mq, o.mq
    scale = 1.0   # Q0 = I and two small rank-1 corrections: the largest element stays ~1
    for r in (0, n // 2 - 256, n - 512):
        assert np.max(np.abs(qg[r:r + 512] - qo[r:r + 512])) <= TOL * scale


def test_ell_n32768_default_queue_run_matches_oracle(gpu, orc):
    """The configuration bench.py times as `n32768-deep` (BASELINE config 4's size on ONE GPU, Q = 8 GiB): 56 deep cuts
    through `queue_run(fused=True)` at the defaults of a new handle of this size -- depth 24, lookahead 32, 48 recorded
    updates per apply pass: groups of 32, 16 on the matrix cores (k_symm_mfma_q2 / k_symm_mfma_q drawing the 64 x 2048 tiles from their queue), the group stage
    sized for 48 slots, the in-run rank-48 apply pass (k_apply_mfma<48> on 512 x 32), a group of 8 left recorded when the
    state is read -- against the oracle's row-parallel loop (tests/test_oracle_pins.py ties it bit for bit to the reference
    loop order): every cut's tsq, xc, kappa, and three 512-row bands of Q (8 GiB each side: bands bound the temporaries)."""
    from ellalgo_rs_amd import synth
    from util import TOL
    n, k = 32768, 56
    kinds, grads, b0, _ = synth.deep_cuts(n, k)
    e = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    assert e.defer_depth == 24 and e.get_option(gpu.capi.OPT_LOOKAHEAD) == 32 and e.get_option(gpu.capi.OPT_QUEUE_DEPTH) == 48
    e.profile_enable(True)
    e.queue_upload(kinds, grads, b0)
    e.queue_run(0, k, fused=True)
    st, ts = e.queue_results()
    prof = e.profile_read()
    assert prof["symv"][1] == 3 and prof["apply"][1] == 1, prof     # groups of 32, 16 | 8; ONE apply pass, at cut 48
    assert np.all(st == 0)
    o = orc.OracleEll.new_with_scalar(1.0, np.zeros(n))
    for i in range(k):
        assert o.update_rowwise_mt(0, grads[i], b0[i], None) == 0
        assert abs(ts[i] - o.tsq) <= TOL * abs(o.tsq), i
    assert abs(e.kappa - o.kappa) <= TOL * abs(o.kappa)
    xo = np.array(o.xc)
    assert np.max(np.abs(e.xc() - xo)) <= TOL * np.max(np.abs(xo))
    qg = e.mq                      # applies the eight recorded updates and mirrors the lower triangle
    qo = o.mq
    scale = float(np.max(np.abs(np.diagonal(qo))))
    for r in (0, n // 2 - 256, n - 512):
        assert np.max(np.abs(qg[r:r + 512] - qo[r:r + 512])) <= TOL * scale, r
        assert np.array_equal(qg[r:r + 512, r:r + 512], qg[r:r + 512, r:r + 512].T)
    # ... and not a comparison of an untouched identity: 56 rank-1 corrections left their mark off the diagonal
    assert np.count_nonzero(qg[n - 512:, :512]) == 512 * 512


def test_ellstable_n32768_persistent_equals_per_block_launches(gpu):
    """n = 32768 is the largest size the persistent solves cover (256 column strips = one workgroup per CU) and the one
    size where k_st_fwd_persist / k_st_bwd_persist are the DEFAULT (the helped forms need two workgroups per strip).
    From a NON-trivial factor (synth.stable_factor: random unit-upper-triangular factor, random diagonal, junk scratch --
    from the identity every off-diagonal entry stays exactly zero and the comparison would be one of zeros): three deep
    cuts; the flag-chained single launches must give the bits of the one-launch-per-block path (same arithmetic, no
    inter-workgroup hand-off) over the whole 8 GiB buffer, compared in bands of rows to bound host memory."""
    from ellalgo_rs_amd import synth
    n, k = 32768, 3
    kinds, grads, b0, _ = synth.deep_cuts(n, k)
    f = synth.stable_factor(n)
    set_default("STABLE_SOLVE", 2)
    a = gpu.EllStable.new_with_matrix(1.0, f, np.zeros(n))
    set_default("STABLE_SOLVE", 0)
    b = gpu.EllStable.new_with_matrix(1.0, f, np.zeros(n))
    set_default("STABLE_SOLVE", 2)
    f0 = f[:8, :64].copy()
    del f
    for i in range(k):
        sa = int(a.update_bias_cut((grads[i], float(b0[i]))))
        sb = int(b.update_bias_cut((grads[i], float(b0[i]))))
        assert sa == sb == 0
        assert a.tsq() == b.tsq() and a.kappa == b.kappa
    assert np.array_equal(a.xc(), b.xc())
    qa = a.mq
    assert not np.array_equal(qa[:8, :64], f0)          # the factor moved ...
    qb = b.mq
    for r in range(0, n, 4096):
        assert np.array_equal(qa[r:r + 4096], qb[r:r + 4096])
    assert np.count_nonzero(np.triu(qa[:4096], 1)) > 4096 * 4096    # ... and the comparison was not one of zeros
    assert np.count_nonzero(np.tril(qa[n - 4096:, n - 4096:], -1)) > 4096 * 1024
