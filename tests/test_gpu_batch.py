"""GPU: the batched small-n engine (include/ellhip_batch.h) against the CPU oracle, through the C ABI.
The engine follows the reference's statement order out of LDS, so the comparison is EXACT (bit for bit), not
the 1e-10 of the large engine."""
import numpy as np
import pytest

from util import mixed_cut

pytestmark = pytest.mark.gpu


def drive(gpu, orc, B, n, K, rounds, seed, *, no_defer=False, nonsym=False, use_parallel=True):
    rng = np.random.default_rng(seed)
    kappa = 0.5 + 2.0 * rng.random(B)
    xc0 = rng.standard_normal((B, n))
    if nonsym:
        mq = np.stack([np.eye(n) * (1.0 + rng.random()) + 0.01 * rng.standard_normal((n, n)) for _ in range(B)])
        batch = gpu.EllBatch.new_with_matrix(kappa, mq, xc0)
        ors = [orc.OracleEll.new_with_matrix(kappa[b], mq[b], xc0[b]) for b in range(B)]
    else:
        batch = gpu.EllBatch.new_with_scalar(kappa, xc0)
        ors = [orc.OracleEll.new_with_scalar(kappa[b], xc0[b]) for b in range(B)]
    if no_defer:
        batch.set_no_defer_trick(True)
        for o in ors:
            o.set_no_defer_trick(True)
    if not use_parallel:
        batch.set_use_parallel_cut(False)
        for o in ors:
            o.set_use_parallel_cut(False)
    counts = np.zeros(4, dtype=int)
    it = 0
    for _ in range(rounds):
        kinds = np.zeros((K, B), dtype=np.int32)
        grads = np.zeros((K, B, n))
        b0 = np.zeros((K, B))
        b1 = np.full((K, B), np.nan)
        want = np.zeros((K, B), dtype=np.int32)
        want_tsq = np.zeros((K, B))
        for k in range(K):
            for b in range(B):
                g = rng.standard_normal(n)
                o = ors[b]
                tau = np.sqrt(max(o.kappa * float(g @ (o.mq @ g)), 0.0))
                kind, c0, c1 = mixed_cut(it + b, g, tau, rng)
                kinds[k, b], grads[k, b], b0[k, b] = kind, g, c0
                if c1 is not None:
                    b1[k, b] = c1
                want[k, b] = o.update(kind, g, c0, c1)
                want_tsq[k, b] = o.tsq
            it += 1
        status, tsq = batch.update(kinds, grads, b0, b1)
        np.testing.assert_array_equal(status, want)
        np.testing.assert_array_equal(tsq, want_tsq)  # NaNs (n = 1: cst1 = inf) compare equal
        counts += np.bincount(status.ravel(), minlength=4)
    np.testing.assert_array_equal(batch.mq, np.stack([o.mq for o in ors]))
    np.testing.assert_array_equal(batch.xc(), np.stack([np.array(o.xc) for o in ors]))
    np.testing.assert_array_equal(batch.kappa, np.array([o.kappa for o in ors]))
    np.testing.assert_array_equal(batch.tsq(), np.array([o.tsq for o in ors]))
    return counts


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 16, 17, 31, 32, 33, 48, 64, 65, 100, 128])
def test_batch_bit_identical_to_oracle(gpu, orc, n):
    B = 7 if n > 32 else 37
    counts = drive(gpu, orc, B, n, K=3, rounds=4 if n > 1 else 1, seed=1000 + n)
    if n > 1:
        assert counts[0] > counts[1:].sum()  # mostly successful cuts, some NoSoln / NoEffect


@pytest.mark.parametrize("n", [4, 16, 40, 128])
def test_batch_no_defer_trick_and_flags(gpu, orc, n):
    drive(gpu, orc, 9, n, K=4, rounds=3, seed=7 + n, no_defer=True)
    drive(gpu, orc, 9, n, K=2, rounds=3, seed=8 + n, use_parallel=False)


@pytest.mark.parametrize("n", [3, 16, 64])
def test_batch_non_symmetric_input_is_mirrored_like_the_reference(gpu, orc, n):
    drive(gpu, orc, 5, n, K=2, rounds=3, seed=50 + n, nonsym=True)


def test_batch_diag_constructor_and_single_cut_per_call(gpu, orc):
    B, n = 11, 6
    rng = np.random.default_rng(3)
    diag = 0.5 + rng.random((B, n))
    xc0 = rng.standard_normal((B, n))
    batch = gpu.EllBatch.new(diag, xc0)
    ors = [orc.OracleEll.new(diag[b], xc0[b]) for b in range(B)]
    assert np.array_equal(batch.mq, np.stack([o.mq for o in ors]))
    g = rng.standard_normal((B, n))
    status, tsq = batch.update(np.zeros(B, dtype=np.int32), g, np.full(B, 0.01))
    for b in range(B):
        assert status[0, b] == ors[b].update_bias_cut(g[b], 0.01)
    assert np.array_equal(batch.mq, np.stack([o.mq for o in ors]))
    batch.set_xc(np.ones((B, n)))
    assert np.array_equal(batch.xc(), np.ones((B, n)))


def test_batch_from_space_clones_an_ell(gpu, orc):
    """BSearchAdaptor pattern (src/cutting_plane.rs:410): B probes start from clones of one space."""
    n, B = 12, 6
    rng = np.random.default_rng(5)
    x0 = rng.standard_normal(n)
    for depth in (1, 8):
        base = gpu.Ell.new_with_scalar(3.0, x0)
        base.defer_depth = depth
        obase = orc.OracleEll.new_with_scalar(3.0, x0)
        for _ in range(3):  # at depth 8 these stay recorded until the clone forces them into Q
            g = rng.standard_normal(n)
            assert int(base.update_bias_cut((g, 0.05))) == obase.update_bias_cut(g, 0.05) == 0
        batch = gpu.EllBatch.from_space(base, B)
        if depth == 1:
            assert np.array_equal(batch.mq[0], base.mq)
        assert np.allclose(batch.mq[B - 1], obase.mq, rtol=1e-12, atol=0)
        assert np.array_equal(batch.xc()[2], base.xc()) and batch.kappa[3] == base.kappa
        # probes diverge from here: different cuts per clone, each equal to a clone of the single space
        probes = [base.clone() for _ in range(B)]
        g = rng.standard_normal((B, n))
        beta = 0.01 * (1 + np.arange(B))
        status, _ = batch.update(np.zeros(B, dtype=np.int32), g, beta)
        for b in range(B):
            assert status[0, b] == int(probes[b].update_bias_cut((g[b], float(beta[b]))))
            assert np.allclose(batch.mq[b], probes[b].mq, rtol=1e-11, atol=1e-300)
            assert np.allclose(batch.xc()[b], probes[b].xc(), rtol=1e-11, atol=1e-300)


def test_batch_argument_checks(gpu):
    with pytest.raises(gpu.capi.EllHipError):
        gpu.EllBatch.new_with_scalar(np.ones(2), np.zeros((2, 129)))
    b = gpu.EllBatch.new_with_scalar(1.0, np.zeros((2, 4)))
    with pytest.raises(ValueError):
        b.update(np.zeros(2, dtype=np.int32), np.zeros((2, 5)), np.zeros(2))
    with pytest.raises(gpu.capi.EllHipError):
        b.update(np.full(2, 7, dtype=np.int32), np.zeros((2, 4)), np.zeros(2))
    st = gpu.EllStable.new_with_scalar(1.0, np.zeros(4))
    with pytest.raises(gpu.capi.EllHipError):
        gpu.EllBatch.from_space(st, 3)


def test_batch_large_population(gpu, orc):
    """20 000 ellipsoids of n = 16: spot-check a sample against the oracle, all statuses Success."""
    B, n, K = 20000, 16, 4
    rng = np.random.default_rng(11)
    batch = gpu.EllBatch.new_with_scalar(1.0, np.zeros((B, n)))
    grads = rng.standard_normal((K, B, n))
    beta = 0.05 * rng.random((K, B))
    status, tsq = batch.update(np.zeros((K, B), dtype=np.int32), grads, beta)
    assert np.all(status == 0)
    mq, xc, kap = batch.mq, batch.xc(), batch.kappa
    for b in rng.choice(B, 25, replace=False):
        o = orc.OracleEll.new_with_scalar(1.0, np.zeros(n))
        for k in range(K):
            assert o.update_bias_cut(grads[k, b], beta[k, b]) == 0
            assert tsq[k, b] == o.tsq
        assert np.array_equal(mq[b], o.mq) and np.array_equal(xc[b], np.array(o.xc)) and kap[b] == o.kappa


def quad_oracle(target):
    """tests/integration_test.rs:85-105 generalised to any n (BASELINE config 1)"""
    def assess(x, gamma):
        d = x - target
        f = 0.0
        for v in d.tolist():  # left fold, as the Rust iterator sum does
            f += v * v
        g = 2.0 * d
        fj = f - gamma
        if fj > 0.0:
            return (g, fj), False, gamma
        return (g, 0.0), True, f
    return assess


def test_config1_n16_problems_side_by_side_equal_the_oracle_runs_exactly(gpu, orc):
    """BASELINE config 1 (n = 16 quadratic, Ell::new_with_scalar(10, 0), Options(2000, 1e-10)) for 12 different
    targets at once: the batched loop makes exactly the decisions of 12 separate oracle-backed loops."""
    n, B, max_iters, tol = 16, 12, 2000, 1e-10
    targets = [np.arange(1, n + 1, dtype=np.float64) + 0.25 * b for b in range(B)]
    # reference loops on the CPU oracle (src/cutting_plane.rs:286-313)
    want = []
    for b in range(B):
        o = orc.OracleEll.new_with_scalar(10.0, np.zeros(n))
        ask = quad_oracle(targets[b])
        gamma, x_best, niter_out = np.inf, None, max_iters
        for niter in range(max_iters):
            x = np.array(o.xc)
            (g, beta), shrunk, gamma = ask(x, gamma)
            if shrunk:
                x_best = x
                st = o.update_central_cut(g, beta)
            else:
                st = o.update_bias_cut(g, beta)
            if st != 0 or o.tsq < tol:
                niter_out = niter
                break
        want.append((x_best, niter_out, gamma))
    # the same loops, all B spaces in one batched handle; finished problems receive a no-op cut (beta = inf: NoSoln)
    batch = gpu.EllBatch.new_with_scalar(10.0, np.zeros((B, n)))
    asks = [quad_oracle(t) for t in targets]
    gamma = [np.inf] * B
    x_best = [None] * B
    niter_out = [max_iters] * B
    done = [False] * B
    for it in range(max_iters):
        if all(done):
            break
        xc = batch.xc()
        kinds = np.zeros(B, dtype=np.int32)
        grads = np.ones((B, n))
        beta = np.full(B, np.inf)
        for b in range(B):
            if done[b]:
                continue
            (g, bt), shrunk, gamma[b] = asks[b](xc[b], gamma[b])
            if shrunk:
                x_best[b] = xc[b].copy()
            kinds[b], grads[b], beta[b] = (1 if shrunk else 0), g, bt
        status, tsq = batch.update(kinds, grads, beta)
        for b in range(B):
            if not done[b] and (status[0, b] != 0 or tsq[0, b] < tol):
                done[b], niter_out[b] = True, it
    for b in range(B):
        xb, ni, gm = want[b]
        assert niter_out[b] == ni and gamma[b] == gm, b
        assert np.array_equal(x_best[b], xb), b
        assert xb is not None and gm < float(np.sum(targets[b] ** 2))  # some progress; the target lies outside the start ball


def test_cpp_batch_mirror_runs_config1_in_bulk(gpu):
    import cpp_build
    exe = cpp_build.build_runner("batch_runner.cpp", "hip")
    res = cpp_build.run_json_lines(exe)
    assert len(res) == 24
    for name, r in res.items():
        # batched engine (bit-exact CPU order) vs one EllHip handle each (1e-10 parity engine) over up to 2000 cuts
        assert r["niter_batch"] == r["niter_single"], r
        # this problem's optimum lies outside the start ball, so 2000 cuts amplify the engines' 1e-16 differences
        assert abs(r["gamma_batch"] - r["gamma_single"]) <= 1e-3 * r["gamma_single"], r
        assert r["max_dx"] < 1.0, r


def test_two_populations_with_different_lds_footprints(gpu, orc):
    """The dynamic-LDS opt-in belongs to the kernel, not to a handle: a population with a large footprint (n = 100)
    must keep launching after a small one (n = 8) has been set up, and the other way round."""
    rng = np.random.default_rng(3)
    big = gpu.EllBatch.new_with_scalar(np.ones(3), np.zeros((3, 100)))
    small = gpu.EllBatch.new_with_scalar(np.ones(5), np.zeros((5, 8)))
    for batch, n, B in ((big, 100, 3), (small, 8, 5), (big, 100, 3)):
        g = rng.standard_normal((1, B, n))
        st, _ = batch.update(np.zeros((1, B), dtype=np.int32), g, np.full((1, B), 0.01))
        assert np.all(st == 0)
    o = orc.OracleEll.new_with_scalar(1.0, np.zeros(8))
    assert np.isfinite(small.tsq()).all() and np.isfinite(big.tsq()).all()


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("n", [3, 16, 40, 100])
def test_random_walk_over_the_batch_api_is_exact(gpu, orc, n, seed):
    """Random sequences of the batched engine's calls -- launches of 1..7 cuts per ellipsoid, centres replaced, the
    two flags toggled between launches -- against one CPU oracle per ellipsoid: every status, tsq and the whole
    state bit for bit."""
    rng = np.random.default_rng(5000 + 37 * seed + n)
    B = int(rng.integers(1, 9))
    kappa = 0.5 + 2.0 * rng.random(B)
    xc0 = rng.standard_normal((B, n))
    batch = gpu.EllBatch.new_with_scalar(kappa, xc0)
    ors = [orc.OracleEll.new_with_scalar(kappa[b], xc0[b]) for b in range(B)]
    it = 0
    ndt = False
    par = True
    for _ in range(12):
        r = rng.random()
        if r < 0.15:
            x = rng.standard_normal((B, n))
            batch.set_xc(x)
            for b, o in enumerate(ors):
                o.set_xc(x[b])
        elif r < 0.25:
            ndt = not ndt
            batch.set_no_defer_trick(ndt)
            for o in ors:
                o.set_no_defer_trick(ndt)
        elif r < 0.32:
            par = not par
            batch.set_use_parallel_cut(par)
            for o in ors:
                o.set_use_parallel_cut(par)
        else:
            K = int(rng.integers(1, 8))
            kinds = np.zeros((K, B), dtype=np.int32)
            grads = np.zeros((K, B, n))
            b0 = np.zeros((K, B))
            b1 = np.full((K, B), np.nan)
            want = np.zeros((K, B), dtype=np.int32)
            want_tsq = np.zeros((K, B))
            for k in range(K):
                for b, o in enumerate(ors):
                    g = rng.standard_normal(n)
                    tau = np.sqrt(max(o.kappa * float(g @ (o.mq @ g)), 0.0))
                    kind, c0, c1 = mixed_cut(it + b, g, tau, rng)
                    kinds[k, b], grads[k, b], b0[k, b] = kind, g, c0
                    if c1 is not None:
                        b1[k, b] = c1
                    want[k, b] = o.update(kind, g, c0, c1)
                    want_tsq[k, b] = o.tsq
                it += 1
            status, tsq = batch.update(kinds, grads, b0, b1)
            np.testing.assert_array_equal(status, want)
            np.testing.assert_array_equal(tsq, want_tsq)
    np.testing.assert_array_equal(batch.mq, np.stack([o.mq for o in ors]))
    np.testing.assert_array_equal(batch.xc(), np.stack([np.array(o.xc) for o in ors]))
    np.testing.assert_array_equal(batch.kappa, np.array([o.kappa for o in ors]))
