"""GPU parity tests for the Ell hot path: the HIP engine (through the C ABI) against the CPU oracle
and against the reference's own one-step known answers (src/ell.rs:236-364)."""
import ctypes as C

import numpy as np
import pytest

from util import assert_state_close, run_mixed, set_default

pytestmark = pytest.mark.gpu


def _approx(a, b, eps=1e-6):  # approx_eq 0.1.8 default, as the reference's assert_approx_eq!
    return abs(a - b) <= eps * max(abs(a), abs(b), 1e-300) or abs(a - b) < eps


# ---------------------------------------------------------------- reference one-step known answers
def test_construct(gpu):  # src/ell.rs:237-244
    e = gpu.Ell.new_with_scalar(0.01, np.zeros(4))
    assert not e.no_defer_trick
    assert _approx(e.kappa, 0.01)
    assert np.array_equal(e.mq, np.eye(4))
    assert np.array_equal(e.xc(), np.zeros(4))
    assert e.tsq() == 0.0


def test_update_central_cut(gpu):  # src/ell.rs:247-256 (xc and mq are assert_eq!: bit-exact)
    e = gpu.Ell.new_with_scalar(0.01, np.zeros(4))
    st = e.update_central_cut((0.5 * np.ones(4), gpu.SingleCut(0.0)))
    assert st == gpu.CutStatus.Success
    assert np.array_equal(e.xc(), -0.01 * np.ones(4))
    assert np.array_equal(e.mq, np.eye(4) - 0.1 * np.ones((4, 4)))
    assert _approx(e.kappa, 0.16 / 15.0)
    assert _approx(e.tsq(), 0.01)


def test_update_bias_cut(gpu):  # src/ell.rs:259-268
    e = gpu.Ell.new_with_scalar(0.01, np.zeros(4))
    st = e.update_bias_cut((0.5 * np.ones(4), gpu.SingleCut(0.05)))
    assert st == gpu.CutStatus.Success
    assert _approx(e.xc()[0], -0.03)
    assert _approx(e.mq[0, 0], 0.8)
    assert _approx(e.kappa, 0.008)
    assert _approx(e.tsq(), 0.01)


def test_update_parallel_central_cut(gpu):  # src/ell.rs:271-280 (bit-exact xc, mq)
    e = gpu.Ell.new_with_scalar(0.01, np.zeros(4))
    st = e.update_central_cut((0.5 * np.ones(4), gpu.ParallelCut(0.0, 0.05)))
    assert st == gpu.CutStatus.Success
    assert np.array_equal(e.xc(), -0.01 * np.ones(4))
    assert np.array_equal(e.mq, np.eye(4) - 0.2 * np.ones((4, 4)))
    assert _approx(e.kappa, 0.012)
    assert _approx(e.tsq(), 0.01)


def test_update_parallel(gpu):  # src/ell.rs:283-292
    e = gpu.Ell.new_with_scalar(0.01, np.zeros(4))
    st = e.update_bias_cut((0.5 * np.ones(4), gpu.ParallelCut(0.01, 0.04)))
    assert st == gpu.CutStatus.Success
    assert _approx(e.xc()[0], -0.0116)
    assert _approx(e.mq[0, 0], 1.0 - 0.232)
    assert _approx(e.kappa, 0.01232)
    assert _approx(e.tsq(), 0.01)


def test_update_parallel_no_effect(gpu):  # src/ell.rs:295-303 (bit-exact xc, mq)
    e = gpu.Ell.new_with_scalar(0.01, np.zeros(4))
    st = e.update_bias_cut((0.5 * np.ones(4), gpu.ParallelCut(-0.04, 0.0625)))
    assert st == gpu.CutStatus.Success
    assert np.array_equal(e.xc(), np.zeros(4))
    assert np.array_equal(e.mq, np.eye(4))
    assert _approx(e.kappa, 0.01)


def test_update_q_no_effect(gpu):  # src/ell.rs:306-314
    e = gpu.Ell.new_with_scalar(0.01, np.zeros(4))
    st = e.update_q((0.5 * np.ones(4), gpu.ParallelCut(-0.04, 0.0625)))
    assert st == gpu.CutStatus.NoEffect
    assert np.array_equal(e.xc(), np.zeros(4))
    assert np.array_equal(e.mq, np.eye(4))
    assert _approx(e.kappa, 0.01)


def test_update_q(gpu):  # src/ell.rs:317-326
    e = gpu.Ell.new_with_scalar(0.01, np.zeros(4))
    st = e.update_q((0.5 * np.ones(4), gpu.ParallelCut(0.01, 0.04)))
    assert st == gpu.CutStatus.Success
    assert _approx(e.xc()[0], -0.0116)
    assert _approx(e.mq[0, 0], 1.0 - 0.232)
    assert _approx(e.kappa, 0.01232)
    assert _approx(e.tsq(), 0.01)


def test_no_defer_trick(gpu):  # src/ell.rs:342-354
    e = gpu.Ell.new_with_scalar(0.01, np.zeros(4))
    e.no_defer_trick = True
    e.update_central_cut((0.5 * np.ones(4), gpu.SingleCut(0.0)))
    assert _approx(e.kappa, 1.0)
    want = (np.eye(4) - 0.1 * np.ones((4, 4))) * (0.16 / 15.0)
    assert np.allclose(e.mq, want, rtol=1e-6, atol=1e-12)


def test_from_covariance(gpu):  # src/ell.rs:357-364
    cov = np.diag([2.0, 3.0, 4.0, 5.0])
    xc = np.array([1.0, 2.0, 3.0, 4.0])
    e = gpu.Ell.from_covariance(cov, xc)
    assert e.kappa == 1.0
    assert np.array_equal(e.mq, cov)
    assert np.array_equal(e.xc(), xc)


def test_new_diag(gpu):  # Ell::new, src/ell.rs:55-57
    e = gpu.Ell.new(np.array([1.0, 2.0, 3.0]), np.array([0.5, 0.0, -0.5]))
    assert e.kappa == 1.0
    assert np.array_equal(e.mq, np.diag([1.0, 2.0, 3.0]))


# ---------------------------------------------------------------- device EllCalc known answers
CALC_CASES = [  # (kind, beta, tsq, status, (rho, sigma, delta))   src/ell_calc.rs:942-1186
    (1, 0.0, 0.01, 0, (0.02, 0.4, 16.0 / 15.0)),
    (0, 0.11, 0.01, 1, None),
    (0, 0.0, 0.01, 0, None),
    (2, -0.05, 0.01, 2, None),
    (0, 0.05, 0.01, 0, (0.06, 0.8, 0.8)),
    (1, (0.0, 0.11), 0.01, 0, (0.02, 0.4, 16.0 / 15.0)),
    (1, (0.0, 0.05), 0.01, 0, (0.02, 0.8, 1.2)),
    (0, (0.07, 0.03), 0.01, 1, None),
    (0, (0.0, 0.05), 0.01, 0, (0.02, 0.8, 1.2)),
    (0, (0.05, 0.11), 0.01, 0, (0.06, 0.8, 0.8)),
    (2, (-0.07, 0.07), 0.01, 2, None),
    (0, (0.01, 0.04), 0.01, 0, (0.0232, 0.928, 1.232)),
    (2, (-0.04, 0.0625), 0.01, 2, None),
    (1, (0.05, None), 0.01, 0, (None, 0.4, None)),
    (0, (0.05, None), 0.01, 0, None),
    (2, (0.05, None), 0.01, 0, None),
    (2, 0.11, 0.01, 1, None),
    (2, 0.01, 0.01, 0, None),
    (2, 0.05, 0.01, 0, (0.06, 0.8, 0.8)),
    (2, (0.07, 0.03), 0.01, 1, None),
    (2, (0.0, 0.05), 0.01, 0, (0.02, 0.8, 1.2)),
    (2, (0.05, 0.11), 0.01, 0, (0.06, 0.8, 0.8)),
    (2, (0.01, 0.04), 0.01, 0, (0.0232, 0.928, 1.232)),
]


@pytest.mark.parametrize("kind,beta,tsq,status,want", CALC_CASES)
def test_device_ellcalc_known_answers(gpu, orc, kind, beta, tsq, status, want):
    st, got = gpu.calc(4, kind, beta, tsq)
    assert int(st) == status
    if want is not None:
        for g, w in zip(got, want):
            if w is not None:
                assert _approx(g, w)
    # and bit-for-bit against the oracle's restatement
    b0, b1 = (beta if isinstance(beta, tuple) else (beta, None))
    so, co = orc.Calc(4).dispatch(kind, b0, b1, tsq)
    assert so == int(st)
    assert tuple(got) == tuple(co)


def test_device_ellcalc_random_bits(gpu, orc):
    """The device coefficient stage must agree with the oracle to the last bit on random inputs
    (same operation order, correctly rounded sqrt / division)."""
    rng = np.random.default_rng(7)
    for n in (2, 3, 16, 4096, 16384):
        calc = orc.Calc(n)
        for _ in range(25):
            tsq = float(10 ** rng.uniform(-6, 2))
            tau = np.sqrt(tsq)
            kind = int(rng.integers(0, 3))
            b0 = float(tau * rng.uniform(-0.5, 1.1))
            b1 = None if rng.random() < 0.4 else float(b0 + tau * rng.uniform(-0.1, 1.5))
            st, got = gpu.calc(n, kind, (b0, b1), tsq)
            so, co = calc.dispatch(kind, b0, b1, tsq)
            assert int(st) == so
            assert tuple(got) == tuple(co), (n, kind, b0, b1, tsq, got, co)


# ---------------------------------------------------------------- sequences against the oracle
@pytest.mark.parametrize("n", [2, 3, 5, 16, 64, 127, 130, 257, 512, 1000, 1024])
def test_mixed_sequence_matches_oracle(gpu, orc, n):
    xc0 = np.linspace(-1.0, 1.0, n)
    g = gpu.Ell.new_with_scalar(2.0, xc0)
    o = orc.OracleEll.new_with_scalar(2.0, xc0)
    k = 48
    nsucc = run_mixed(g, o, k, seed=100 + n, check_every=8)
    assert nsucc >= k // 2
    assert_state_close(g, o, what=f"n={n} final")


@pytest.mark.parametrize("n", [2048, 4096, 8192])
def test_deep_cuts_large_matches_oracle(gpu, orc, n):
    """Config-2 style deep cuts at a size where a Q row spans many wave steps."""
    from ellalgo_rs_amd import synth
    kinds, grads, b0, _ = synth.deep_cuts(n, 6)   # n = 8192 exercises the padded leading dimension + nt policy
    g = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    g.defer_depth = 1                             # the reference's data flow (n = 8192 would start at depth 16)
    o = orc.OracleEll.new_with_scalar(1.0, np.zeros(n))
    for i in range(6):
        so = o.update(0, grads[i], b0[i])
        sg = g.update_bias_cut((grads[i], float(b0[i])))
        assert int(sg) == so == 0
    assert_state_close(g, o, what=f"n={n}")


def test_no_defer_trick_sequence(gpu, orc):
    n = 96
    g = gpu.Ell.new_with_scalar(3.0, np.ones(n))
    o = orc.OracleEll.new_with_scalar(3.0, np.ones(n))
    g.no_defer_trick = True
    o.set_no_defer_trick(True)
    run_mixed(g, o, 40, seed=5, check_every=10)
    assert g.kappa == 1.0 or g.kappa == o.kappa
    assert_state_close(g, o, what="no_defer")


def test_rank1_bit_identical_for_same_gt(gpu, orc):
    """With Q0 = I the GEMV is exact (gt = g), so the whole update must be BIT-identical to the
    reference loop order: this pins the (ratio*gt[hi])*gt[lo] identity and the absence of FMA."""
    n = 200
    rng = np.random.default_rng(3)
    gr = rng.standard_normal(n)
    g = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    o = orc.OracleEll.new_with_scalar(1.0, np.zeros(n))
    assert g.update_bias_cut((gr, 0.1)) == 0 and o.update(0, gr, 0.1) == 0
    # omega differs only by the summation tree; everything downstream is elementwise
    if g.tsq() == o.tsq:
        assert np.array_equal(g.mq, o.mq)
        assert np.array_equal(g.xc(), o.xc)
    else:
        assert_state_close(g, o, tol=1e-14)
    assert np.array_equal(g.mq, g.mq.T)  # symmetric to the bit


def test_failed_cut_leaves_state_untouched(gpu, orc):  # src/ell.rs:105-109
    n = 33
    g = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    gr = np.ones(n)
    q0, x0, k0 = g.mq, g.xc(), g.kappa
    st = g.update_bias_cut((gr, 100.0))  # tsq = n < beta^2
    assert st == gpu.CutStatus.NoSoln
    assert g.tsq() == float(n)           # tsq IS updated
    assert np.array_equal(g.mq, q0) and np.array_equal(g.xc(), x0) and g.kappa == k0
    st = g.update_central_cut((gr, (0.0, -1.0)))  # beta1 < 0
    assert st == gpu.CutStatus.NoSoln
    assert np.array_equal(g.mq, q0)


def test_degenerate_zero_gradient_nan(gpu, orc):
    """benches/ellipsoid.rs starts at xc = 0 with g = 2*xc = 0: omega = 0, the update 'succeeds'
    and fills the state with NaN.  The engine must reproduce that, not guard it."""
    n = 10
    g = gpu.Ell.new_with_scalar(10.0, np.zeros(n))
    o = orc.OracleEll.new_with_scalar(10.0, np.zeros(n))
    sg = g.update_bias_cut((np.zeros(n), 0.0))
    so = o.update(0, np.zeros(n), 0.0)
    assert int(sg) == so == 0
    assert g.tsq() == 0.0 and o.tsq == 0.0
    assert np.isnan(g.kappa) and np.isnan(o.kappa)
    assert np.isnan(g.xc()).all() and np.isnan(g.mq).all()


def test_dimension_one_infinite_cst1(gpu, orc):
    """n = 1: cst1 = n^2/(n^2-1) = inf (src/ell_calc.rs:67); kappa becomes inf, as in the reference."""
    g = gpu.Ell.new_with_scalar(1.0, np.zeros(1))
    o = orc.OracleEll.new_with_scalar(1.0, np.zeros(1))
    sg = g.update_central_cut((np.array([2.0]), 0.0))
    so = o.update(1, np.array([2.0]), 0.0)
    assert int(sg) == so == 0
    np.testing.assert_array_equal(g.mq, o.mq)
    np.testing.assert_array_equal(g.xc(), o.xc)
    assert g.kappa == o.kappa == np.inf


def test_clone_is_independent(gpu, orc):  # Clone, used by BSearchAdaptor (src/cutting_plane.rs:410)
    n = 40
    a = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    rng = np.random.default_rng(1)
    a.update_bias_cut((rng.standard_normal(n), 0.01))
    b = a.clone()
    assert np.array_equal(a.mq, b.mq) and a.kappa == b.kappa and a.tsq() == b.tsq()
    qa = a.mq
    b.update_central_cut((rng.standard_normal(n), 0.0))
    assert np.array_equal(a.mq, qa)
    assert not np.array_equal(b.mq, qa)
    b.set_xc(np.full(n, 7.0))
    assert np.array_equal(b.xc(), np.full(n, 7.0))
    assert not np.array_equal(a.xc(), b.xc())


def test_nonsymmetric_input_is_mirrored_like_the_reference(gpu, orc):  # src/ell.rs:124-126
    n = 70
    rng = np.random.default_rng(11)
    a = rng.standard_normal((n, n)) * 0.01
    mq = np.eye(n) + a  # NOT symmetric
    g = gpu.Ell.new_with_matrix(1.0, mq, np.zeros(n))
    o = orc.OracleEll.new_with_matrix(1.0, mq, np.zeros(n))
    # a failing cut first: nothing may be mirrored yet
    gr = rng.standard_normal(n)
    assert g.update_bias_cut((gr, 1e6)) == 1 and o.update(0, gr, 1e6) == 1
    assert np.array_equal(g.mq, mq)
    for i in range(5):
        gr = rng.standard_normal(n)
        assert int(g.update_bias_cut((gr, 0.01))) == o.update(0, gr, 0.01) == 0
        assert_state_close(g, o, what=f"nonsym step {i}")
    assert np.array_equal(g.mq, g.mq.T)


def test_queue_matches_direct_updates(gpu, orc):
    set_default("RESIDENT", 0)   # the STREAMED schedules are compared bit for bit here; the resident queue run sums Q g in its own shape (test_gpu_resident.py)
    from ellalgo_rs_amd import synth
    n, k = 384, 12
    kinds, grads, b0, b1 = synth.parallel_cuts(n, k)
    a = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    b = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    o = orc.OracleEll.new_with_scalar(1.0, np.zeros(n))
    a.queue_upload(kinds, grads, b0, b1)
    a.queue_run(0, k)
    st, ts = a.queue_results()
    for i in range(k):
        sb = b._update(int(kinds[i]), (grads[i], (b0[i], b1[i])))
        so = o.update(int(kinds[i]), grads[i], b0[i], b1[i])
        assert int(sb) == so == st[i] == 0
        assert ts[i] == b.tsq()
    assert np.array_equal(a.mq, b.mq) and np.array_equal(a.xc(), b.xc()) and a.kappa == b.kappa
    assert_state_close(a, o, what="queue")


def test_queue_halts_at_first_failure(gpu):
    set_default("RESIDENT", 0)   # the STREAMED schedules are compared bit for bit here; the resident queue run sums Q g in its own shape (test_gpu_resident.py)
    n, k = 64, 6
    rng = np.random.default_rng(2)
    grads = rng.standard_normal((k, n))
    kinds = np.zeros(k, dtype=np.int32)
    b0 = np.array([0.01, 0.01, 1e9, 0.01, 0.01, 0.01])
    a = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    b = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    a.queue_upload(kinds, grads, b0)
    a.queue_run(0, k)
    st, _ = a.queue_results()
    assert list(st) == [0, 0, 1, 3, 3, 3]
    for i in range(2):
        b.update_bias_cut((grads[i], b0[i]))
    assert np.array_equal(a.mq, b.mq) and a.kappa == b.kappa
    # the handle is usable again afterwards
    assert a.update_bias_cut((grads[3], 0.01)) == 0


def test_row_sharded_two_phase_equals_unsharded(gpu, orc):
    """The multi-GPU schedule on one card: two row-block shards share one gt buffer (the in-place
    all-gather), each runs phase 1, both run the redundant scalar stage and their own rank-1."""
    pkg = gpu
    L = pkg.capi.load()
    n, half = 256, 128
    rng = np.random.default_rng(9)
    ref = pkg.Ell.new_with_scalar(1.5, np.zeros(n))
    hs = []
    for r in range(2):
        h = C.c_void_p()
        pkg.capi.check(L.ellhip_create_shard(C.byref(h), n, r * half, half, 1.5, None, None, None, -1))
        hs.append(h)
    pkg.capi.check(L.ellhip_set_gt_dev(hs[1], L.ellhip_gt_dev(hs[0]), None))
    try:
        for i in range(10):
            g = np.ascontiguousarray(rng.standard_normal(n))
            gp = g.ctypes.data_as(C.c_void_p)
            b0 = 0.02
            for h in hs:
                pkg.capi.check(L.ellhip_update_begin(h, 0, gp, b0, 0, 0.0))
            for h in hs:
                pkg.capi.check(L.ellhip_synchronize(h))   # "all-gather complete"
            sts = [pkg.capi.check(L.ellhip_update_end(h)) for h in hs]
            assert sts == [0, 0]
            assert int(ref.update_bias_cut((g, b0))) == 0
        q = np.empty((n, n))
        for r, h in enumerate(hs):
            blk = np.empty((half, n))
            pkg.capi.check(L.ellhip_get_mq(h, blk.ctypes.data_as(C.c_void_p)))
            q[r * half:(r + 1) * half] = blk
            x = np.empty(n)
            pkg.capi.check(L.ellhip_get_xc(h, x.ctypes.data_as(C.c_void_p)))
            assert np.array_equal(x, ref.xc())             # every rank holds the same xc
            assert L.ellhip_kappa(h) == ref.kappa
        assert np.array_equal(q, ref.mq)                    # bit-identical to the unsharded engine
    finally:
        for h in hs:
            L.ellhip_destroy(h)


def test_bad_arguments_return_error_codes(gpu):
    L = gpu.capi.load()
    h = C.c_void_p()
    assert L.ellhip_create(C.byref(h), 0, 0, 1.0, None, None, None, -1) == gpu.capi.E_INVALID
    assert L.ellhip_create(C.byref(h), 7, 4, 1.0, None, None, None, -1) == gpu.capi.E_INVALID
    e = gpu.Ell.new_with_scalar(1.0, np.zeros(4))
    assert L.ellhip_update(e._h, 5, np.zeros(4).ctypes.data_as(C.c_void_p), 0.0, 0, 0.0) == gpu.capi.E_INVALID
    assert L.ellhip_update(e._h, 0, None, 0.0, 0, 0.0) == gpu.capi.E_INVALID
    assert L.ellhip_update_end(e._h) == gpu.capi.E_STATE
    with pytest.raises(ValueError):
        e.update_bias_cut((np.zeros(5), 0.0))   # n mismatch: the reference would panic (src/arr.rs:427-429)


# ---------------------------------------------------------------- pipelined (one pass per update)
def _pipelined_run(space, cuts):
    """prime / cut / commit driver in the shape of include/ellhip.h; returns statuses and tsqs."""
    out = []
    space.prime(cuts[0][1])
    for i, (kind, g, b0, b1) in enumerate(cuts):
        st = space.cut(kind, (b0, b1))
        out.append((int(st), space.tsq(), space.kappa, space.xc()))
        nxt = cuts[i + 1][1] if i + 1 < len(cuts) else None
        space.commit(nxt)
    return out


@pytest.mark.parametrize("n,no_defer", [(2, False), (33, False), (256, False), (1000, True), (2048, False)])
def test_pipelined_is_bit_identical_to_update(gpu, orc, n, no_defer):
    """ellhip_prime/cut/commit (shrink of cut k fused with the GEMV of cut k+1) must give exactly the
    bits of ellhip_update, including across failed cuts (which skip the shrink but not the GEMV)."""
    from util import mixed_cut
    rng = np.random.default_rng(77 + n)
    a = gpu.Ell.new_with_scalar(2.0, np.linspace(-1, 1, n))
    b = gpu.Ell.new_with_scalar(2.0, np.linspace(-1, 1, n))
    o = orc.OracleEll.new_with_scalar(2.0, np.linspace(-1, 1, n))
    a.no_defer_trick = b.no_defer_trick = no_defer
    o.set_no_defer_trick(no_defer)
    cuts, ref = [], []
    for i in range(24):
        g = rng.standard_normal(n)
        g /= np.linalg.norm(g)
        tau = float(np.sqrt(max(o.kappa * (g @ (o.mq @ g)), 0.0)))
        kind, b0, b1 = mixed_cut(i, g, tau, rng)
        cuts.append((kind, g, b0, b1))
        o.update(kind, g, b0, b1)
        st = a._update(kind, (g, (b0, b1)))
        ref.append((int(st), a.tsq(), a.kappa, a.xc()))
    got = _pipelined_run(b, cuts)
    assert any(r[0] != 0 for r in ref) and any(r[0] == 0 for r in ref)
    for i, (r, g_) in enumerate(zip(ref, got)):
        assert r[0] == g_[0] and r[1] == g_[1] and r[2] == g_[2], f"step {i}: {r[:3]} vs {g_[:3]}"
        assert np.array_equal(r[3], g_[3]), f"xc step {i}"
    assert np.array_equal(a.mq, b.mq)
    assert_state_close(b, o, what=f"pipelined n={n}")


def test_pipelined_observers_commit_pending_shrink(gpu):
    """get_mq / clone / update in the middle of a pipelined sequence see the shrunk Q."""
    n = 96
    rng = np.random.default_rng(8)
    g1, g2 = rng.standard_normal(n), rng.standard_normal(n)
    a = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    b = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    assert a.update_bias_cut((g1, 0.01)) == 0
    b.prime(g1)
    assert b.cut(0, 0.01) == 0
    assert np.array_equal(a.mq, b.mq)             # get_mq committed the shrink
    c = b.clone()
    assert np.array_equal(c.mq, a.mq)
    assert a.update_central_cut((g2, 0.0)) == 0 and b.update_central_cut((g2, 0.0)) == 0
    assert np.array_equal(a.mq, b.mq) and a.kappa == b.kappa


def test_pipelined_nonsymmetric_input(gpu, orc):
    n = 50
    rng = np.random.default_rng(12)
    mq = np.eye(n) + 0.01 * rng.standard_normal((n, n))
    cuts = [(0, rng.standard_normal(n), 0.01, None) for _ in range(4)]
    a = gpu.Ell.new_with_matrix(1.0, mq, np.zeros(n))
    b = gpu.Ell.new_with_matrix(1.0, mq, np.zeros(n))
    for kind, g, b0, b1 in cuts:
        assert a._update(kind, (g, b0)) == 0
    _pipelined_run(b, cuts)
    assert np.array_equal(a.mq, b.mq) and np.array_equal(a.xc(), b.xc())


@pytest.mark.parametrize("n", [384, 8192])
def test_queue_fused_is_bit_identical_to_queue(gpu, n):
    set_default("RESIDENT", 0)   # the STREAMED schedules are compared bit for bit here; the resident queue run sums Q g in its own shape (test_gpu_resident.py)
    set_default("LOOKAHEAD", 3)  # ... and the matrix-core groups of the default lookahead theirs (test_gpu_overlap.py); up to 3 queued cuts per pass keep k_symv's arithmetic
    from ellalgo_rs_amd import synth
    k = 10
    kinds, grads, b0, b1 = synth.parallel_cuts(n, k)
    a = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    b = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    a.queue_upload(kinds, grads, b0, b1)
    b.queue_upload(kinds, grads, b0, b1)
    a.queue_run(0, k)
    b.queue_run(0, 3, fused=True)       # split like the benchmark: the second run finds cut 3 primed
    b.queue_run(3, k - 3, fused=True)
    sa, ta = a.queue_results()
    sb, tb = b.queue_results()
    assert np.array_equal(sa, sb) and np.array_equal(ta, tb) and np.all(sa == 0)
    assert np.array_equal(a.mq, b.mq) and np.array_equal(a.xc(), b.xc()) and a.kappa == b.kappa


def test_queue_fused_halts_at_first_failure(gpu):
    set_default("RESIDENT", 0)   # the STREAMED schedules are compared bit for bit here; the resident queue run sums Q g in its own shape (test_gpu_resident.py)
    n, k = 64, 6
    rng = np.random.default_rng(2)
    grads = rng.standard_normal((k, n))
    kinds = np.zeros(k, dtype=np.int32)
    b0 = np.array([0.01, 0.01, 1e9, 0.01, 0.01, 0.01])
    a = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    b = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    a.queue_upload(kinds, grads, b0)
    a.queue_run(0, k, fused=True)
    st, _ = a.queue_results()
    assert list(st) == [0, 0, 1, 3, 3, 3]
    for i in range(2):
        b.update_bias_cut((grads[i], b0[i]))
    assert np.array_equal(a.mq, b.mq) and a.kappa == b.kappa
    assert a.update_bias_cut((grads[3], 0.01)) == 0 and b.update_bias_cut((grads[3], 0.01)) == 0
    assert np.array_equal(a.mq, b.mq)
