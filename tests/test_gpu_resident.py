"""GPU: queue runs with the matrix parked on-chip (csrc/resident_kernels.hpp: one persistent launch per batch of cuts,
the lower triangle in the register files, one grid barrier per cut) -- what `ellhip_queue_run` / `_run_fused` choose for
an unsharded Ell handle with n <= 4224 and at least 4 cuts (ELLHIP_OPT_RESIDENT, default 1).  Restates the loop
src/cutting_plane.rs:299-311 around Ell::update_core (src/ell.rs:97-137).  Against the CPU oracle to the north-star
tolerance, against the streamed schedules to rounding, bit-reproducible run to run; all three super-tile sizes (R = 1, 2,
3), odd and ragged n, all six EllCalc entry points, a failing cut, batches mixed with streamed updates and observers."""
import numpy as np
import pytest

from util import TOL, assert_state_close, dense_spd, set_default

pytestmark = pytest.mark.gpu


def _cuts(n, k, seed, fail_at=None):
    """k cuts over all six EllCalc entry points with betas scaled to tau ~ 1 (Q0 = I, kappa0 = 1, |g| = 1)."""
    rng = np.random.default_rng(seed)
    grads = rng.standard_normal((k, n))
    grads /= np.linalg.norm(grads, axis=1)[:, None]
    kinds = np.zeros(k, dtype=np.int32)
    b0 = np.zeros(k)
    b1 = np.full(k, np.nan)
    for i in range(k):
        m = i % 6
        if m == 0:
            kinds[i], b0[i] = 0, 0.02 * rng.random()                                   # update_bias_cut(SingleCut)
        elif m == 1:
            kinds[i], b0[i] = 1, 0.0                                                   # update_central_cut(SingleCut)
        elif m == 2:
            kinds[i], b0[i], b1[i] = 0, 0.01 * rng.random(), 0.1 + 0.1 * rng.random()  # update_bias_cut(ParallelCut)
        elif m == 3:
            kinds[i], b0[i], b1[i] = 1, 0.0, 0.05 + 0.1 * rng.random()                 # update_central_cut(ParallelCut)
        elif m == 4:
            kinds[i], b0[i] = 2, 0.01 * rng.random()                                   # update_q(SingleCut)
        else:
            kinds[i], b0[i], b1[i] = 2, 0.005 * rng.random(), 0.1 + 0.1 * rng.random()  # update_q(ParallelCut)
    if fail_at is not None:
        kinds[fail_at], b0[fail_at], b1[fail_at] = 0, 1e6, np.nan                      # NoSoln
    return kinds, grads, b0, b1


def _resident_launches(e):
    return e.profile_read()["resident"][1]


@pytest.mark.parametrize("n,dense", [(2, 1), (63, 1), (64, 1), (65, 1), (130, 1), (1000, 0), (1001, 0), (1408, 0), (1409, 0), (2048, 0),
                                     (2817, 0), (2818, 0), (4096, 0), (4224, 0), (1408, 2), (2048, 2), (2817, 2), (4096, 2)])
def test_resident_queue_run_matches_the_oracle(gpu, orc, n, dense):
    """One batch of 24 cuts from a non-trivial start (xc != 0, kappa != 1; dense = 1: a random SPD matrix for the small
    sizes, dense = 2: a dense SPD matrix at R = 1, 2 and 3 -- n = 4096: the LDS-resident third tile column holds data
    at entry, not zeros)."""
    k = 24
    kinds, grads, b0, b1 = _cuts(n, k, 100 + n)
    xc0 = np.linspace(-1.0, 1.0, n)
    if dense == 2:
        q0 = dense_spd(n, 7 * n)
        e = gpu.Ell.new_with_matrix(1.5, q0, xc0)
        o = orc.OracleEll.new_with_matrix(1.5, q0, xc0)
        scale = float(np.sqrt(1.5))
        b0, b1 = b0 * scale, b1 * scale
        del q0
    elif n <= 130:
        rng = np.random.default_rng(n)
        a = rng.standard_normal((n, n)) * 0.1
        q0 = np.eye(n) + a @ a.T
        q0 = 0.5 * (q0 + q0.T)            # symmetric to the bit (a non-symmetric input takes the mirror path first)
        e = gpu.Ell.new_with_matrix(1.5, q0, xc0)
        o = orc.OracleEll.new_with_matrix(1.5, q0, xc0)
        scale = float(np.sqrt(1.5 * np.max(np.linalg.eigvalsh(q0))))
        b0, b1 = b0 * scale, b1 * scale
    else:
        e = gpu.Ell.new_with_scalar(1.0, xc0)
        o = orc.OracleEll.new_with_scalar(1.0, xc0)
    e.profile_enable(True)
    e.queue_upload(kinds, grads, b0, b1)
    e.queue_run(0, k, fused=(n % 2 == 0))
    st, ts = e.queue_results()
    assert _resident_launches(e) == 1, "the batch did not take the resident kernel"
    nsucc = 0
    for i in range(k):
        so = o.update(int(kinds[i]), grads[i], b0[i], None if np.isnan(b1[i]) else b1[i])
        assert int(st[i]) == so, (i, int(st[i]), so)
        assert abs(ts[i] - o.tsq) <= TOL * abs(o.tsq), i
        if so != 0:                       # (tiny n: the ellipsoid collapses within the batch) the queue halts there
            assert all(int(x) == 3 for x in st[i + 1:])
            break
        nsucc += 1
    assert nsucc == k or n < 64
    assert_state_close(e, o, what=f"resident n={n}")
    q = e.mq
    assert np.array_equal(q, q.T)      # the mirrored half is rebuilt from the lower triangle the kernel wrote


@pytest.mark.parametrize("n,depth", [(192, 1), (1024, 1), (1024, 8), (2112, 16), (4096, 8)])
def test_resident_batches_mixed_with_streamed_updates_and_observers(gpu, orc, n, depth):
    """direct updates -> resident batch -> direct updates (recorded, not yet applied at depth 8 / 16) -> resident batch
    (applies them first) -> clone -> a streamed pipelined batch on the clone (option off) and a resident one on the
    original: every hand-over between the schedules keeps the state the oracle has."""
    set_default("SYMV_MIN_N", 512)
    k = 40
    kinds, grads, b0, b1 = _cuts(n, k, 7 * n + depth)
    xc0 = np.zeros(n)
    e = gpu.Ell.new_with_scalar(1.0, xc0)
    e.defer_depth = depth
    o = orc.OracleEll.new_with_scalar(1.0, xc0)
    e.profile_enable(True)

    def beta(i):
        return (b0[i], None if np.isnan(b1[i]) else b1[i])

    def oracle_to(j0, j1):
        for i in range(j0, j1):
            assert o.update(int(kinds[i]), grads[i], b0[i], None if np.isnan(b1[i]) else b1[i]) == 0

    for i in range(0, 5):
        assert int(e._update(int(kinds[i]), (grads[i], beta(i)))) == 0
    e.queue_upload(kinds, grads, b0, b1)
    e.queue_run(5, 9, fused=True)
    for i in range(14, 17):
        assert int(e._update(int(kinds[i]), (grads[i], beta(i)))) == 0
    e.queue_run(17, 7, fused=False)
    oracle_to(0, 24)
    assert abs(e.tsq() - o.tsq) <= TOL * abs(o.tsq) and abs(e.kappa - o.kappa) <= TOL * abs(o.kappa)
    c = e.clone()
    assert_state_close(c, o, what="clone after two resident batches")
    c.set_option(gpu.capi.OPT_RESIDENT, 0)
    c.queue_upload(kinds, grads, b0, b1)
    c.queue_run(24, 16, fused=True)
    e.queue_run(24, 16, fused=True)
    st_c, ts_c = c.queue_results()
    st_e, ts_e = e.queue_results()
    assert _resident_launches(e) == 3
    oracle_to(24, 40)
    assert np.all(st_c[24:] == 0) and np.all(st_e[5:14] == 0) and np.all(st_e[17:] == 0)
    assert np.max(np.abs(ts_c[24:] - ts_e[24:]) / ts_e[24:]) <= 1e-12      # streamed vs resident: to rounding
    assert_state_close(e, o, what="resident")
    assert_state_close(c, o, what="streamed clone")
    assert np.max(np.abs(e.mq - c.mq)) <= 1e-12 * np.max(np.abs(c.mq))


@pytest.mark.parametrize("n", [200, 2048, 3000])
def test_failing_cut_halts_the_resident_batch(gpu, orc, n):
    """A NoSoln cut in the middle: Q, xc, kappa untouched by it, every later cut reports ELLHIP_UNKNOWN, the queue stays
    halted for a following run until its results are read, and direct updates work again afterwards
    (src/cutting_plane.rs:308; the same contract as the streamed queue)."""
    k, bad = 20, 11
    kinds, grads, b0, b1 = _cuts(n, k, 5 * n, fail_at=bad)
    e = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    o = orc.OracleEll.new_with_scalar(1.0, np.zeros(n))
    e.queue_upload(kinds, grads, b0, b1)
    e.queue_run(0, 14, fused=True)
    e.queue_run(14, 6, fused=True)          # still halted: nothing runs
    st, ts = e.queue_results()
    for i in range(bad):
        assert o.update(int(kinds[i]), grads[i], b0[i], None if np.isnan(b1[i]) else b1[i]) == 0
    assert list(st[:bad]) == [0] * bad and int(st[bad]) == 1 and all(int(x) == 3 for x in st[bad + 1:])
    so = o.update(int(kinds[bad]), grads[bad], b0[bad], None)
    assert so == 1 and abs(ts[bad] - o.tsq) <= TOL * abs(o.tsq)
    assert_state_close(e, o, what="after the halt")
    i = bad + 1
    assert int(e._update(int(kinds[i]), (grads[i], (b0[i], None if np.isnan(b1[i]) else b1[i])))) == \
        o.update(int(kinds[i]), grads[i], b0[i], None if np.isnan(b1[i]) else b1[i]) == 0
    assert_state_close(e, o, what="direct update after the halt")


def test_resident_runs_are_bit_reproducible_and_batching_does_not_matter(gpu):
    """Static assignment, fixed summation orders, no atomics in the data path: the same bits run after run, and whether
    the cuts arrive as one batch or several."""
    n, k = 2817, 36
    kinds, grads, b0, b1 = _cuts(n, k, 99)
    outs = []
    for pieces in ([(0, 36)], [(0, 36)], [(0, 5), (5, 17), (22, 14)]):
        e = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
        e.queue_upload(kinds, grads, b0, b1)
        for a, c in pieces:
            e.queue_run(a, c, fused=True)
        st, ts = e.queue_results()
        assert np.all(st == 0)
        outs.append((ts, e.xc(), e.kappa, e.mq))
    for other in outs[1:]:
        assert np.array_equal(outs[0][0], other[0]) and np.array_equal(outs[0][1], other[1]) and outs[0][2] == other[2]
        assert np.array_equal(outs[0][3], other[3])


def test_what_does_not_take_the_resident_kernel(gpu):
    """Short runs, larger matrices, no_defer_trick, the option switched off, EllStable: the streamed schedules."""
    from ellalgo_rs_amd import synth
    kinds, grads, b0, b1 = synth.deep_cuts(4288, 8)
    e = gpu.Ell.new_with_scalar(1.0, np.zeros(4288))     # 67 tile rows: 23 super-tile rows at R = 3, more than 256 CUs hold
    e.profile_enable(True)
    e.queue_upload(kinds, grads, b0, b1)
    e.queue_run(0, 8, fused=True)
    e.synchronize()
    assert _resident_launches(e) == 0
    kinds, grads, b0, b1 = synth.deep_cuts(512, 12)
    e = gpu.Ell.new_with_scalar(1.0, np.zeros(512))
    e.profile_enable(True)
    e.queue_upload(kinds, grads, b0, b1)
    e.queue_run(0, 3, fused=True)                         # fewer than 4 cuts
    e.no_defer_trick = True
    e.queue_run(3, 4, fused=True)
    e.no_defer_trick = False
    e.set_option(gpu.capi.OPT_RESIDENT, 0)
    e.queue_run(7, 4, fused=False)
    e.synchronize()
    assert _resident_launches(e) == 0
    e.set_option(gpu.capi.OPT_RESIDENT, 1)
    e.queue_run(8, 4, fused=False)
    e.synchronize()
    assert _resident_launches(e) == 1
    s = gpu.EllStable.new_with_scalar(1.0, np.zeros(512))
    with pytest.raises(gpu.capi.EllHipError):
        s.set_option(gpu.capi.OPT_RESIDENT, 1)


@pytest.mark.parametrize("n,fault_at", [(200, 0), (2048, 5), (4096, 11), (4096, 23)])
def test_abandoned_resident_batch_is_undone_and_rerun_on_the_streamed_schedule(gpu, orc, n, fault_at):
    """A bounded in-launch wait that gives up (here: one workgroup abandons the batch at cut `fault_at`, ELLHIP_OPT_RESIDENT_FAULT)
    must never leave a half-updated matrix: no workgroup writes its tiles back (rs_commit), xc / the scalar state / the
    batch's queue results are restored, and the same call reruns the batch on the streamed schedule.  Checked against a
    twin handle that took the streamed schedule for that batch from the start -- to the bit, which it can only be when the
    pre-batch state was restored exactly -- and against the oracle."""
    k = 36
    kinds, grads, b0, b1 = _cuts(n, k, 31 * n + fault_at)
    twins = []
    for faulty in (True, False):
        e = gpu.Ell.new_with_scalar(1.0, np.linspace(-1.0, 1.0, n))
        e.profile_enable(True)
        e.queue_upload(kinds, grads, b0, b1)
        e.queue_run(0, 12, fused=True)                       # a resident batch that commits
        assert e.get_option(gpu.capi.OPT_RESIDENT_ABANDONED) == 0
        if faulty:
            e.set_option(gpu.capi.OPT_RESIDENT_FAULT, fault_at)
        else:
            e.set_option(gpu.capi.OPT_RESIDENT, 0)
        e.queue_run(12, 24, fused=True)                      # abandoned at cut 12 + fault_at, undone, rerun | streamed
        if faulty:
            assert e.get_option(gpu.capi.OPT_RESIDENT_ABANDONED) == 1 and e.get_option(gpu.capi.OPT_RESIDENT) == 0
            assert _resident_launches(e) == 2
        st, ts = e.queue_results()
        assert np.all(st == 0)
        twins.append((ts, e.xc(), e.kappa, e.mq, e))
    a, b = twins
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2] and np.array_equal(a[3], b[3])
    o = orc.OracleEll.new_with_scalar(1.0, np.linspace(-1.0, 1.0, n))
    for i in range(k):
        assert o.update_rowwise_mt(int(kinds[i]), grads[i], b0[i], None if np.isnan(b1[i]) else b1[i]) == 0
        assert abs(a[0][i] - o.tsq) <= TOL * abs(o.tsq), i
    assert_state_close(a[4], o, what=f"after an abandoned batch n={n}")
    # the handle keeps working (streamed from now on), and switching the option back on brings the resident kernel back
    e = a[4]
    e.set_option(gpu.capi.OPT_RESIDENT_FAULT, -1)
    e.set_option(gpu.capi.OPT_RESIDENT, 1)
    e.queue_upload(kinds, grads, b0, b1)
    e.queue_run(0, 4, fused=True)
    e.queue_results()
    assert _resident_launches(e) == 1 and e.get_option(gpu.capi.OPT_RESIDENT_ABANDONED) == 1


def test_abandoned_batch_with_a_failing_cut_inside(gpu, orc):
    """The batch is abandoned AFTER one of its cuts failed?  No: the kernel leaves the loop at the failing cut, so a fault
    placed behind it never fires and the batch commits; a fault in front of it abandons the batch and the rerun halts at
    the same cut.  Both end in the oracle's state with the same queue results."""
    n, k, bad = 2048, 20, 9
    kinds, grads, b0, b1 = _cuts(n, k, 4242, fail_at=bad)
    outs = []
    for fault_at, abandoned in ((15, 0), (3, 1)):
        e = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
        e.set_option(gpu.capi.OPT_RESIDENT_FAULT, fault_at)
        e.queue_upload(kinds, grads, b0, b1)
        e.queue_run(0, k, fused=False)
        st, ts = e.queue_results()
        assert e.get_option(gpu.capi.OPT_RESIDENT_ABANDONED) == abandoned
        assert list(st[:bad]) == [0] * bad and int(st[bad]) == 1 and all(int(x) == 3 for x in st[bad + 1:])
        outs.append(e)
    o = orc.OracleEll.new_with_scalar(1.0, np.zeros(n))
    for i in range(bad):
        assert o.update(int(kinds[i]), grads[i], b0[i], None if np.isnan(b1[i]) else b1[i]) == 0
    assert o.update(int(kinds[bad]), grads[bad], b0[bad], None) == 1
    for e in outs:
        assert_state_close(e, o, what="failing cut inside a batch")


def test_two_handles_run_resident_batches_from_two_host_threads(gpu, orc):
    """BSearchAdaptor clones the space per probe (src/cutting_plane.rs:409-418): two handles on one card, each driven from its
    own host thread through resident batches of n = 4096 (253 workgroups each: two such grids cannot be co-resident).  The
    launches are cooperative, so the runtime runs one grid at a time; whatever the interleaving, every handle must end in
    the oracle's state -- right results, or a batch abandoned and rerun (counted), never a matrix that mixes update counts."""
    import threading
    n, k, piece = 4096, 96, 12
    kinds, grads, b0, b1 = _cuts(n, k, 909)
    hs = []
    for _ in range(2):
        e = gpu.Ell.new_with_scalar(1.0, np.linspace(-1.0, 1.0, n))
        e.queue_upload(kinds, grads, b0, b1)
        hs.append(e)
    start = threading.Barrier(2)
    errs = []

    def worker(e):
        try:
            start.wait()
            for a in range(0, k, piece):
                e.queue_run(a, piece, fused=True)
        except Exception as ex:   # noqa: BLE001 -- reported below
            errs.append(repr(ex))

    ts = [threading.Thread(target=worker, args=(e,)) for e in hs]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    o = orc.OracleEll.new_with_scalar(1.0, np.linspace(-1.0, 1.0, n))
    want = np.empty(k)
    for i in range(k):
        assert o.update_rowwise_mt(int(kinds[i]), grads[i], b0[i], None if np.isnan(b1[i]) else b1[i]) == 0
        want[i] = o.tsq
    for e in hs:
        st, tsq = e.queue_results()
        assert np.all(st == 0)
        assert np.max(np.abs(tsq - want) / want) <= TOL
        assert_state_close(e, o, what=f"concurrent resident batches (abandoned: {e.get_option(gpu.capi.OPT_RESIDENT_ABANDONED)})")
