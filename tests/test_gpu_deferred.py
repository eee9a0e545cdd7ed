"""GPU: deferred shrink (ellhip_set_defer_depth(8)): cuts are recorded as (sigma/omega, gt) pairs, GEMVs are
corrected with the recorded pairs, one pass applies 8 of them.  Must stay within the 1e-10 parity
tolerance of the CPU oracle, and be bit-identical across schedules and row partitions for that depth."""
import ctypes as C

import numpy as np
import pytest

from util import assert_state_close, run_mixed, set_default

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [2, 3, 5, 16, 64, 127, 130, 257, 1000, 1024, 2048])
def test_deferred_mixed_sequence_matches_oracle(gpu, orc, n):
    xc0 = np.linspace(-1.0, 1.0, n)
    g = gpu.Ell.new_with_scalar(2.0, xc0)
    g.defer_depth = 8
    assert g.defer_depth == 8
    o = orc.OracleEll.new_with_scalar(2.0, xc0)
    nsucc = run_mixed(g, o, 52, seed=500 + n, check_every=5)   # 5 is coprime to 8: get_mq flushes at every phase
    assert nsucc >= 26
    assert_state_close(g, o, what=f"deferred n={n}")


@pytest.mark.parametrize("n", [512, 1000, 2112, 4096])
def test_deferred_symv_mixed_sequence_matches_oracle(gpu, orc, n, monkeypatch):
    """Depth 8 with the lower-triangle GEMV forced on at small sizes (default threshold n >= 5120)."""
    set_default("SYMV_MIN_N", 512)
    xc0 = np.linspace(-1.0, 1.0, n)
    g = gpu.Ell.new_with_scalar(2.0, xc0)
    g.defer_depth = 8
    o = orc.OracleEll.new_with_scalar(2.0, xc0)
    nsucc = run_mixed(g, o, 44, seed=900 + n, check_every=11)
    assert nsucc >= 22
    assert_state_close(g, o, what=f"deferred+symv n={n}")


@pytest.mark.parametrize("n", [4096, 8192])
def test_deferred_deep_cuts_large(gpu, orc, n):
    from ellalgo_rs_amd import synth
    k = 11
    kinds, grads, b0, _ = synth.deep_cuts(n, k)
    g = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    g.defer_depth = 8
    o = orc.OracleEll.new_with_scalar(1.0, np.zeros(n))
    for i in range(k):
        assert int(g.update_bias_cut((grads[i], float(b0[i])))) == o.update(0, grads[i], b0[i]) == 0
        assert abs(g.tsq() - o.tsq) <= 1e-10 * abs(o.tsq)
    assert_state_close(g, o, what=f"deferred n={n}")


def test_deferred_all_schedules_bit_identical(gpu):
    """direct updates == two-pass queue == pipelined queue == prime/cut/commit, for depth 8."""
    set_default("RESIDENT", 0)   # the STREAMED schedules are compared bit for bit here; the resident queue run sums Q g in its own shape (test_gpu_resident.py)
    from ellalgo_rs_amd import synth
    n, k = 640, 21
    kinds, grads, b0, b1 = synth.parallel_cuts(n, k)
    spaces = [gpu.Ell.new_with_scalar(1.0, np.zeros(n)) for _ in range(4)]
    for s in spaces:
        s.defer_depth = 8
    a, b, c, d = spaces
    for i in range(k):
        assert int(a._update(int(kinds[i]), (grads[i], (b0[i], b1[i])))) == 0
    b.queue_upload(kinds, grads, b0, b1)
    b.queue_run(0, k)
    c.queue_upload(kinds, grads, b0, b1)
    c.queue_run(0, 5, fused=True)
    c.queue_run(5, k - 5, fused=True)
    d.prime(grads[0])
    for i in range(k):
        assert int(d.cut(int(kinds[i]), (b0[i], b1[i]))) == 0
        d.commit(grads[i + 1] if i + 1 < k else None)
    sb, tb = b.queue_results()
    sc, tc = c.queue_results()
    assert np.all(sb == 0) and np.array_equal(sb, sc) and np.array_equal(tb, tc)
    qa = a.mq
    for s in (b, c, d):
        assert s.kappa == a.kappa and s.tsq() == a.tsq()
        assert np.array_equal(s.xc(), a.xc())
        assert np.array_equal(s.mq, qa)
    assert np.array_equal(qa, qa.T)


def test_deferred_failed_cuts_record_nothing(gpu, orc):
    n = 48
    rng = np.random.default_rng(21)
    g = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    g.defer_depth = 8
    o = orc.OracleEll.new_with_scalar(1.0, np.zeros(n))
    for i in range(30):
        gr = rng.standard_normal(n)
        beta = 1e6 if i % 4 == 3 else 0.01     # every 4th cut fails
        assert int(g.update_bias_cut((gr, beta))) == o.update(0, gr, beta)
    assert_state_close(g, o, what="deferred with failures")


def test_deferred_queue_halts_and_recovers(gpu, orc):
    n, k = 64, 14
    rng = np.random.default_rng(22)
    grads = rng.standard_normal((k, n))
    kinds = np.zeros(k, dtype=np.int32)
    b0 = np.full(k, 0.01)
    b0[10] = 1e9
    g = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    g.defer_depth = 8
    o = orc.OracleEll.new_with_scalar(1.0, np.zeros(n))
    g.queue_upload(kinds, grads, b0)
    g.queue_run(0, k, fused=True)
    st, _ = g.queue_results()
    assert list(st) == [0] * 10 + [1] + [3] * 3
    for i in range(10):
        o.update(0, grads[i], b0[i])
    o.update(0, grads[10], b0[10])
    assert_state_close(g, o, what="after halt")
    assert int(g.update_bias_cut((grads[11], 0.01))) == o.update(0, grads[11], 0.01) == 0
    assert_state_close(g, o, what="after recovery")


def test_deferred_clone_and_mode_switches(gpu, orc):
    n = 80
    rng = np.random.default_rng(23)
    g = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    g.defer_depth = 8
    o = orc.OracleEll.new_with_scalar(1.0, np.zeros(n))
    for i in range(5):
        gr = rng.standard_normal(n)
        g.update_bias_cut((gr, 0.01)), o.update(0, gr, 0.01)
    c = g.clone()                      # flushes the source, copies the depth
    assert c.defer_depth == 8
    assert_state_close(c, o, what="clone")
    g.defer_depth = 1                  # back to the immediate data flow
    g.no_defer_trick = True
    o.set_no_defer_trick(True)
    for i in range(3):
        gr = rng.standard_normal(n)
        g.update_central_cut((gr, 0.0)), o.update(1, gr, 0.0)
    assert_state_close(g, o, what="after mode switches")


def test_deferred_row_shards_bit_identical_to_one_shard_and_close_to_unsharded(gpu, monkeypatch):
    """Row partitions agree bit for bit with each other (2 shards == 1 shard holding all rows: both use the
    full-row GEMV).  The unsharded handle takes the lower-triangle GEMV at depth 8, so it matches to
    rounding, not to the bit."""
    pkg = gpu
    L = pkg.capi.load()
    n, half = 1024, 512
    rng = np.random.default_rng(9)
    set_default("SYMV_MIN_N", 512)
    ref = pkg.Ell.new_with_scalar(1.5, np.zeros(n))
    ref.defer_depth = 8

    def shard(row0, nrows):
        h = C.c_void_p()
        pkg.capi.check(L.ellhip_create_shard(C.byref(h), n, row0, nrows, 1.5, None, None, None, -1))
        pkg.capi.check(L.ellhip_set_defer_depth(h, 8))
        return h

    hs = [shard(0, half), shard(half, half)]
    one = shard(0, n)
    pkg.capi.check(L.ellhip_set_gt_dev(hs[1], L.ellhip_gt_dev(hs[0]), None))
    try:
        for i in range(19):
            g = np.ascontiguousarray(rng.standard_normal(n))
            gp = g.ctypes.data_as(C.c_void_p)
            for h in hs:
                pkg.capi.check(L.ellhip_update_begin(h, 0, gp, 0.02, 0, 0.0))
            for h in hs:
                pkg.capi.check(L.ellhip_synchronize(h))
            assert [pkg.capi.check(L.ellhip_update_end(h)) for h in hs] == [0, 0]
            assert pkg.capi.check(L.ellhip_update(one, 0, gp, 0.02, 0, 0.0)) == 0
            assert int(ref.update_bias_cut((g, 0.02))) == 0
        q = np.empty((n, n))
        q1 = np.empty((n, n))
        pkg.capi.check(L.ellhip_get_mq(one, q1.ctypes.data_as(C.c_void_p)))
        x1 = np.empty(n)
        pkg.capi.check(L.ellhip_get_xc(one, x1.ctypes.data_as(C.c_void_p)))
        for r, h in enumerate(hs):
            blk = np.empty((half, n))
            pkg.capi.check(L.ellhip_get_mq(h, blk.ctypes.data_as(C.c_void_p)))
            q[r * half:(r + 1) * half] = blk
            x = np.empty(n)
            pkg.capi.check(L.ellhip_get_xc(h, x.ctypes.data_as(C.c_void_p)))
            assert np.array_equal(x, x1) and L.ellhip_kappa(h) == L.ellhip_kappa(one)
        assert np.array_equal(q, q1)                                   # partitions: bit-identical
        qr = ref.mq
        assert np.max(np.abs(q - qr)) <= 1e-12 * np.max(np.abs(qr))    # vs the lower-triangle GEMV: rounding only
        assert np.max(np.abs(x1 - ref.xc())) <= 1e-12 * np.max(np.abs(x1))
        assert abs(L.ellhip_kappa(one) - ref.kappa) <= 1e-13 * abs(ref.kappa)
    finally:
        for h in hs + [one]:
            L.ellhip_destroy(h)


def test_symv_equals_full_gemv_to_rounding(gpu, monkeypatch):
    """The lower-triangle GEMV (4 n^2 bytes) against the full-row GEMV on the same deferred sequence."""
    from ellalgo_rs_amd import synth
    n, k = 2048 + 64, 20          # not a multiple of the segment width: exercises the ragged diagonal segment
    kinds, grads, b0, b1 = synth.parallel_cuts(n, k)
    set_default("SYMV_MIN_N", 512)   # the default threshold (8192) would skip it at this size
    a = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    a.defer_depth = 8
    set_default("SYMV", 0)
    b = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    b.defer_depth = 8
    set_default("SYMV", 1)
    for i in range(k):
        cut = (grads[i], (b0[i], b1[i]))
        assert int(a._update(int(kinds[i]), cut)) == int(b._update(int(kinds[i]), cut)) == 0
        assert abs(a.tsq() - b.tsq()) <= 1e-13 * abs(b.tsq())
    qa, qb = a.mq, b.mq
    assert np.array_equal(qa, qa.T)
    assert np.max(np.abs(qa - qb)) <= 1e-12 * np.max(np.abs(qb))
    assert np.max(np.abs(a.xc() - b.xc())) <= 1e-12 * np.max(np.abs(b.xc()))


@pytest.mark.parametrize("n", [512, 1000, 2112, 4096])
@pytest.mark.parametrize("depth", [8, 16, 24])
def test_lower_triangle_schedule_depths_match_oracle(gpu, orc, n, depth, monkeypatch):
    """k_symv + k_apply_lower (16-row tiles) at depth 8 and 16, forced on at small sizes; get_mq in between
    exercises the mirror at every phase of the pending count."""
    set_default("SYMV_MIN_N", 512)
    xc0 = np.linspace(-1.0, 1.0, n)
    g = gpu.Ell.new_with_scalar(2.0, xc0)
    g.defer_depth = depth
    assert g.defer_depth == depth
    o = orc.OracleEll.new_with_scalar(2.0, xc0)
    nsucc = run_mixed(g, o, 60, seed=1300 + n + depth, check_every=7)
    assert nsucc >= 30
    assert_state_close(g, o, what=f"lower schedule depth {depth} n={n}")


def test_depth16_needs_the_lower_triangle_schedule(gpu):
    g = gpu.Ell.new_with_scalar(1.0, np.zeros(1024))   # below the default threshold (8192)
    with pytest.raises(gpu.capi.EllHipError):
        g.defer_depth = 16
    g = gpu.Ell.new_with_scalar(1.0, np.zeros(1001))
    with pytest.raises(gpu.capi.EllHipError):
        g.defer_depth = 16


def test_apply_kernels_agree_bit_for_bit(gpu, monkeypatch):
    """k_apply_lower (16-row tiles) and k_sweep_apply<LOWER> (4-row tiles) put every lower-triangle element through
    the same roundings."""
    set_default("SYMV_MIN_N", 512)
    n, k = 1536, 24
    from ellalgo_rs_amd import synth
    kinds, grads, b0, b1 = synth.parallel_cuts(n, k)
    outs = []
    for kern in ("0", "1"):
        set_default("APPLY_KERNEL", int(kern))
        e = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
        e.defer_depth = 8
        e.queue_upload(kinds, grads, b0, b1)
        e.queue_run(0, k)
        st, _ = e.queue_results()
        assert np.all(st == 0)
        outs.append((e.mq, e.xc(), e.kappa))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]


def test_default_depth_of_new_handles(gpu, monkeypatch):
    """ellhip_create: depth 24 wherever the lower-triangle schedule exists (unsharded Ell, even n >= 5120), depth 8
    for other n >= 3072, else the reference's data flow; ELLHIP_OPT_AUTO_DEFER = 0 keeps depth 1 everywhere; clones
    inherit; the setter overrides."""
    assert gpu.Ell.new_with_scalar(1.0, np.zeros(2048)).defer_depth == 1
    assert gpu.Ell.new_with_scalar(1.0, np.zeros(4096)).defer_depth == 8      # 3072 <= n < 5120: full-row GEMVs, depth 8
    assert gpu.Ell.new_with_scalar(1.0, np.zeros(5118)).defer_depth == 8
    assert gpu.Ell.new_with_scalar(1.0, np.zeros(5120)).defer_depth == 24     # the threshold (tools/midsize_sweep.py)
    assert gpu.Ell.new_with_scalar(1.0, np.zeros(8191)).defer_depth == 8      # odd n: no 16-byte pairs, no lower schedule
    e = gpu.Ell.new_with_scalar(1.0, np.zeros(8192))
    assert e.defer_depth == 24 and e.clone().defer_depth == 24
    e.defer_depth = 1
    assert e.defer_depth == 1 and e.clone().defer_depth == 1
    set_default("AUTO_DEFER", 0)
    assert gpu.Ell.new_with_scalar(1.0, np.zeros(8192)).defer_depth == 1
    set_default("AUTO_DEFER", 1)
    # a caller-supplied NON-symmetric matrix: the first successful update mirrors it as the reference does, then
    # the recorded schedule takes over -- same state as depth 1 to rounding
    n = 8192
    rng = np.random.default_rng(11)
    mq = np.eye(n)
    mq[5, 3] = 0.25          # lower-triangle entry without its mirror image
    from ellalgo_rs_amd import synth
    kinds, grads, b0, _ = synth.deep_cuts(n, 3)
    a = gpu.Ell.new_with_matrix(1.0, mq, np.zeros(n))
    b = gpu.Ell.new_with_matrix(1.0, mq, np.zeros(n))
    b.defer_depth = 1
    assert a.defer_depth == 24
    for i in range(3):
        assert int(a.update_bias_cut((grads[i], float(b0[i])))) == int(b.update_bias_cut((grads[i], float(b0[i])))) == 0
    qa, qb = a.mq, b.mq
    assert qa[3, 5] == qa[5, 3] and np.max(np.abs(qa - qb)) <= 1e-12 * np.max(np.abs(qb))
    assert np.max(np.abs(a.xc() - b.xc())) <= 1e-12 * np.max(np.abs(b.xc()))


@pytest.mark.parametrize("depth", [8, 16, 24])
def test_observers_between_prime_and_cut_keep_the_primed_gradient_valid(gpu, orc, depth, monkeypatch):
    """A gradient that is primed but not yet cut carries y = Q_base*g for the base the recorded updates belong to.
    ellhip_flush / get_mq / clone in that window apply the recorded updates (Q_base changes), so the library must
    recompute y: the sequence has to come out the same as the undisturbed one."""
    from ellalgo_rs_amd import synth
    from util import TOL
    set_default("SYMV_MIN_N", 512)
    n, k = 1024, 30
    kinds, grads, b0, b1 = synth.parallel_cuts(n, k)
    a = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    a.defer_depth = depth
    a.queue_upload(kinds, grads, b0, b1)
    a.queue_run(0, k, fused=True)
    sa, ta = a.queue_results()
    assert np.all(sa == 0)
    # queue path: the fused run leaves the next cut primed; flush / get_mq / clone in between
    b = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    b.defer_depth = depth
    b.queue_upload(kinds, grads, b0, b1)
    b.queue_run(0, 5, fused=True)
    b.flush()
    b.queue_run(5, 6, fused=True)
    q_mid = b.mq
    b.queue_run(11, 7, fused=True)
    c = b.clone()
    b.queue_run(18, k - 18, fused=True)
    sb, tb = b.queue_results()
    assert np.all(sb == 0)
    assert np.max(np.abs(ta - tb) / np.abs(ta)) <= 1e-12
    assert np.max(np.abs(a.mq - b.mq)) <= 1e-12 * np.max(np.abs(a.mq))
    assert np.max(np.abs(a.xc() - b.xc())) <= 1e-12 * np.max(np.abs(a.xc()))
    # direct path: prime / cut / commit(next) with observers while `next` is primed
    d = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    d.defer_depth = depth
    d.prime(grads[0])
    for i in range(k):
        beta = (b0[i], None if np.isnan(b1[i]) else b1[i])
        assert int(d.cut(int(kinds[i]), beta)) == 0
        assert abs(d.tsq() - ta[i]) <= 1e-12 * abs(ta[i]), f"cut {i}"
        d.commit(grads[i + 1] if i + 1 < k else None)
        if i == 4:
            d.flush()
        if i == 10:
            assert np.max(np.abs(d.mq - q_mid)) <= 1e-12 * np.max(np.abs(q_mid))
        if i == 17:
            e = d.clone()
            assert np.max(np.abs(e.mq - c.mq)) <= 1e-12 * np.max(np.abs(c.mq))
    assert np.max(np.abs(a.mq - d.mq)) <= 1e-12 * np.max(np.abs(a.mq))
    # and the oracle agrees with all of them
    o = orc.OracleEll.new_with_scalar(1.0, np.zeros(n))
    for i in range(k):
        assert o.update(int(kinds[i]), grads[i], b0[i], None if np.isnan(b1[i]) else b1[i]) == 0
    for sp in (a, b, d):
        assert np.max(np.abs(sp.mq - o.mq)) <= TOL * np.max(np.abs(o.mq))
        assert np.max(np.abs(sp.xc() - o.xc)) <= TOL * np.max(np.abs(o.xc))


def test_observers_on_a_halted_queue_see_the_recorded_updates(gpu, orc):
    """A queue halts at its first failing cut; the updates recorded before it belong to successful cuts and must
    reach Q for whoever looks (get_mq, clone, a depth switch) even before ellhip_queue_results has been read."""
    n, k = 64, 14
    rng = np.random.default_rng(29)
    grads = rng.standard_normal((k, n))
    kinds = np.zeros(k, dtype=np.int32)
    b0 = np.full(k, 0.01)
    b0[10] = 1e9
    g = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    g.defer_depth = 8
    o = orc.OracleEll.new_with_scalar(1.0, np.zeros(n))
    for i in range(11):
        o.update(0, grads[i], b0[i])
    g.queue_upload(kinds, grads, b0)
    g.queue_run(0, k, fused=True)
    assert np.max(np.abs(g.mq - o.mq)) <= 1e-10 * np.max(np.abs(o.mq))     # 2 updates were still recorded
    c = g.clone()
    g.defer_depth = 1
    st, _ = g.queue_results()
    assert list(st) == [0] * 10 + [1] + [3] * 3
    assert_state_close(g, o, what="halted queue, depth switched")
    assert_state_close(c, o, what="clone of a halted queue")
    for sp in (g, c):
        assert int(sp.update_bias_cut((grads[11], 0.01))) == 0
    assert o.update(0, grads[11], 0.01) == 0
    assert_state_close(g, o, what="after recovery (depth 1)")
    assert_state_close(c, o, what="after recovery (clone, depth 8)")


@pytest.mark.parametrize("depth", [1, 8, 16])
def test_long_run_stays_inside_the_parity_tolerance(gpu, orc, depth, monkeypatch):
    """SURVEY 8d asks for parity after 1, 10 and 100 updates; this one runs 600 deep cuts (n = 1536, every schedule
    incl. the lower-triangle one) and checks the state against the oracle at 100, 300 and 600: the rounding
    differences of the GEMV orders do not accumulate beyond the 1e-10 tolerance."""
    from ellalgo_rs_amd import synth
    set_default("SYMV_MIN_N", 512)
    n, k = 1536, 600
    kinds, grads, b0, b1 = synth.deep_cuts(n, k)
    b0 = 0.3 * b0      # (the synthetic betas are sized for 220 cuts: keep tau above them for 600)
    e = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    e.defer_depth = depth
    e.queue_upload(kinds, grads, b0, b1)
    o = orc.OracleEll.new_with_scalar(1.0, np.zeros(n))
    done = 0
    for upto in (100, 300, 600):
        e.queue_run(done, upto - done, fused=True)
        for i in range(done, upto):
            assert o.update(0, grads[i], b0[i]) == 0     # (single-threaded literal loop: ~10 ms per update here)
        done = upto
        assert_state_close(e, o, what=f"depth {depth} after {upto} updates")
    st, ts = e.queue_results()
    assert np.all(st == 0)


@pytest.mark.parametrize("n,depth", [(40, 8), (640, 8), (640, 16), (640, 24)])
def test_depth_switches_between_prime_and_cut_do_not_leak_dot_products(gpu, orc, n, depth, monkeypatch):
    """A prime on a recorded schedule leaves the scalar stage's dot products behind for the NEXT cut.  If the depth is
    switched to 1 before that cut, a fused pass primes the following gradient without any, and the depth is switched
    back, the old ones must not be taken for the new gradient's."""
    set_default("SYMV_MIN_N", 512)
    rng = np.random.default_rng(77)
    gs = [rng.standard_normal(n) for _ in range(6)]
    e = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    o = orc.OracleEll.new_with_scalar(1.0, np.zeros(n))
    e.defer_depth = depth
    for g in gs[:3]:                                   # a few recorded updates first
        assert int(e.update_bias_cut((g, 0.01))) == o.update(0, g, 0.01) == 0
    e.prime(gs[3])                                     # dot products of gs[3] are in place
    e.defer_depth = 1                                  # (applies the recorded updates, re-primes)
    assert int(e.cut(0, (0.01, None))) == o.update(0, gs[3], 0.01) == 0
    e.commit(gs[4])                                    # fused rank-1 + GEMV primes gs[4]: no dot products
    e.defer_depth = depth
    assert int(e.cut(0, (0.01, None))) == o.update(0, gs[4], 0.01) == 0
    assert abs(e.tsq() - o.tsq) <= 1e-10 * abs(o.tsq)
    e.commit(gs[5])
    assert int(e.cut(0, (0.01, None))) == o.update(0, gs[5], 0.01) == 0
    e.commit(None)
    assert_state_close(e, o, what=f"n={n} depth {depth}")


@pytest.mark.parametrize("n", [5, 130, 257, 1000, 4096])
def test_dot_products_beside_the_full_row_gemv_equal_the_separate_launch(gpu, n):
    """ELLHIP_OPT_FUSE_DOTS (default 1): on the recorded full-row schedule the v_j . g partial sums are formed by extra
    workgroups of the GEMV's launch (k_sweep_gemv_dots) and g . y inside k_scalar_apply_def -- in k_scalar_dot_def's exact
    shape, so the bits equal those of the separate launch (option 0): direct updates, the two-pass queue and the
    pipelined queue, with a failing cut."""
    set_default("RESIDENT", 0)   # the STREAMED schedules are compared bit for bit here; the resident queue run sums Q g in its own shape (test_gpu_resident.py)
    from ellalgo_rs_amd import synth
    k = 21
    kinds, grads, b0, b1 = synth.parallel_cuts(n, k)
    b0 = b0.copy()
    b0[13] = 1e6   # fails: NoSoln halts the queues there

    def build(flag):
        s = gpu.Ell.new_with_scalar(1.0, np.linspace(-1.0, 1.0, n))
        s.defer_depth = 8
        s.set_option(gpu.capi.OPT_FUSE_DOTS, flag)
        assert s.get_option(gpu.capi.OPT_FUSE_DOTS) == flag
        return s

    ref, direct, twopass, piped = build(0), build(1), build(1), build(1)
    for i in range(13):
        cut = (grads[i], (b0[i], b1[i]))
        assert int(ref._update(int(kinds[i]), cut)) == int(direct._update(int(kinds[i]), cut)) == 0
        assert ref.tsq() == direct.tsq() and ref.kappa == direct.kappa
    for s, fused in ((twopass, False), (piped, True)):
        s.queue_upload(kinds, grads, b0, b1)
        s.queue_run(0, k, fused=fused)
        st, _ = s.queue_results()
        assert list(st[:14]) == [0] * 13 + [1]
    for s in (direct, twopass, piped):
        assert np.array_equal(s.xc(), ref.xc()) and s.kappa == ref.kappa
        assert np.array_equal(s.mq, ref.mq)


@pytest.mark.parametrize("n,depth", [(1024, 8), (1024, 16), (2112, 16), (2112, 24)])
def test_dot_products_from_the_symv_reduction_agree_with_the_separate_launch(gpu, n, depth):
    """ELLHIP_OPT_FUSE_DOTS on the lower-triangle schedule: k_symv_reduce<NP> yields g . y and v_j . g beside y (partial
    sums per 128 columns); with the option off the separate k_scalar_dot_def launch forms them in its own shape.  The two
    associate the dot products differently, so they agree to rounding (1e-13), not to the bit; the drivers of ONE form
    (direct updates, two-pass queue, pipelined queue, with a failing cut) agree bit for bit."""
    set_default("RESIDENT", 0)   # the STREAMED schedules are compared bit for bit here; the resident queue run sums Q g in its own shape (test_gpu_resident.py)
    set_default("LOOKAHEAD", 3)  # ... and groups of up to 3 queued cuts keep k_symv's arithmetic (the matrix-core groups of the default, 12, agree to rounding: test_gpu_overlap.py)
    from ellalgo_rs_amd import synth
    set_default("SYMV_MIN_N", 512)
    k = 37
    kinds, grads, b0, b1 = synth.parallel_cuts(n, k)
    b0 = b0.copy()
    b0[29] = 1e6   # fails: the queues halt there

    def build(flag):
        s = gpu.Ell.new_with_scalar(1.0, np.linspace(-1.0, 1.0, n))
        s.defer_depth = depth
        s.set_option(gpu.capi.OPT_FUSE_DOTS, flag)
        assert s.defer_depth == depth
        return s

    ref, direct, twopass, piped = build(0), build(1), build(1), build(1)
    for i in range(29):
        cut = (grads[i], (b0[i], b1[i]))
        assert int(ref._update(int(kinds[i]), cut)) == int(direct._update(int(kinds[i]), cut)) == 0
        assert abs(ref.tsq() - direct.tsq()) <= 1e-13 * ref.tsq() and abs(ref.kappa - direct.kappa) <= 1e-13 * ref.kappa
    for s, fused in ((twopass, False), (piped, True)):
        s.queue_upload(kinds, grads, b0, b1)
        s.queue_run(0, 11, fused=fused)
        s.queue_run(11, k - 11, fused=fused)
        st, _ = s.queue_results()
        assert list(st[:30]) == [0] * 29 + [1]
    qd = direct.mq
    for s in (twopass, piped):
        assert np.array_equal(s.xc(), direct.xc()) and s.kappa == direct.kappa
        assert np.array_equal(s.mq, qd)
    assert np.max(np.abs(ref.xc() - direct.xc())) <= 1e-12 * np.max(np.abs(ref.xc()))
    assert np.max(np.abs(ref.mq - qd)) <= 1e-12 * np.max(np.abs(qd))


@pytest.mark.parametrize("n,depth", [(512, 8), (1000, 16), (2112, 24), (4096 + 64, 24), (8192, 16)])
def test_matrix_core_apply_pass_matches_the_oracle(gpu, orc, n, depth):
    """ELLHIP_OPT_APPLY_KERNEL = 2: the recorded updates applied as one rank-NP update on the FP64 matrix cores
    (k_apply_mfma).  One rounding per update and element instead of the reference's two, so it is compared with the
    oracle (1e-10) and with k_apply_lower (to rounding), not bit for bit; ragged last strips and column blocks included."""
    set_default("SYMV_MIN_N", 512)
    set_default("RESIDENT", 0)
    xc0 = np.linspace(-1.0, 1.0, n)
    outs = []
    for kern in (2, 1):
        g = gpu.Ell.new_with_scalar(2.0, xc0)
        g.defer_depth = depth
        g.set_option(gpu.capi.OPT_APPLY_KERNEL, kern)
        assert g.get_option(gpu.capi.OPT_APPLY_KERNEL) == kern and g.defer_depth == depth
        assert g.get_option(gpu.capi.OPT_QUEUE_DEPTH) == 48   # (round 3: the four schedule keys fell through to this one)
        g.profile_enable(True)
        o = orc.OracleEll.new_with_scalar(2.0, xc0)
        nsucc = run_mixed(g, o, 3 * depth + 5, seed=4100 + n + depth, check_every=depth + 3)
        assert nsucc >= depth + 4
        assert g.profile_read()["apply"][1] >= 1   # at least one apply pass of the pinned kernel ran before the observer's
        assert_state_close(g, o, what=f"apply kernel {kern} depth {depth} n={n}")
        outs.append((g.mq, g.xc(), g.kappa))
    assert np.max(np.abs(outs[0][0] - outs[1][0])) <= 1e-12 * np.max(np.abs(outs[1][0]))
    assert np.array_equal(outs[0][0], outs[0][0].T)
    assert np.max(np.abs(outs[0][1] - outs[1][1])) <= 1e-12 * np.max(np.abs(outs[1][1]))


def test_every_per_handle_option_round_trips(gpu):
    """ellhip_set_option -> ellhip_get_option for every per-handle key, and nothing else moves (round 3: SYMV, SYMV_MIN_N,
    APPLY_LOWER and APPLY_KERNEL were silently stored into QUEUE_DEPTH, unvalidated)."""
    capi = gpu.capi
    ell_keys = {"SYMV": [0, 1], "SYMV_MIN_N": [512, 4096, 5120], "APPLY_LOWER": [0, 1], "APPLY_KERNEL": [0, 1, 2, -1],
                "FUSE_DOTS": [0, 1], "RESIDENT": [0, 1], "OVERLAP": [0, 1], "LOOKAHEAD": [1, 3, 16, 32], "QUEUE_DEPTH": [0, 48]}
    e = gpu.Ell.new_with_scalar(1.0, np.zeros(640))
    snapshot = lambda s, keys: {k: s.get_option(getattr(capi, "OPT_" + k)) for k in keys}
    for name, values in ell_keys.items():
        for v in values:
            before = snapshot(e, ell_keys)
            e.set_option(getattr(capi, "OPT_" + name), v)
            after = snapshot(e, ell_keys)
            before[name] = v
            assert after == before, (name, v)
    for name, bad in (("QUEUE_DEPTH", 4096), ("LOOKAHEAD", 33), ("APPLY_KERNEL", 3), ("SYMV_MIN_N", 8), ("SYMV", 2)):
        with pytest.raises(Exception):
            e.set_option(getattr(capi, "OPT_" + name), bad)
    # the handle still works, on the schedule the options describe
    e.set_option(capi.OPT_SYMV_MIN_N, 512)
    e.defer_depth = 24
    assert e.defer_depth == 24
    e.set_option(capi.OPT_SYMV, 0)      # the lower-triangle schedule is gone: depth 24 falls back to 8
    assert e.defer_depth == 8
    st_keys = {"STABLE_SOLVE": [0, 1, 2, 3], "STABLE_FACTOR": [0, 1, 2]}
    s = gpu.EllStable.new_with_scalar(1.0, np.zeros(64))
    for name, values in st_keys.items():
        for v in values:
            before = snapshot(s, st_keys)
            s.set_option(getattr(capi, "OPT_" + name), v)
            before[name] = v
            assert snapshot(s, st_keys) == before
