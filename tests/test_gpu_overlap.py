"""GPU: pipelined queue runs on the lower-triangle schedule that use what a QUEUE knows -- the next gradients.
The GEMV y = Q_base g of a queued cut reads a matrix the cuts before it do not change until the next apply pass (recorded
schedule: src/ell.rs:117-128 is deferred), so
  * ELLHIP_OPT_LOOKAHEAD = L (default 32; csrc/ellhip_capi.hip queue_run_multi): the products of L consecutive queued cuts
    are formed in ONE pass over the lower triangle -- L <= 3 on the vector ALU (k_symv_multi, per vector k_symv's
    arithmetic), L > 3 on the FP64 matrix cores (k_symm_mfma, n a multiple of 64; own association: a few ulp) with the
    group's scalar stage in four launches (group_kernels.hpp) and, at depth 24, up to ELLHIP_OPT_QUEUE_DEPTH = 48
    recorded updates per apply pass inside a run;
  * ELLHIP_OPT_OVERLAP (default 1, used where LOOKAHEAD is 1; queue_run_overlapped): the next cut's GEMV is issued on a
    second stream beside this cut's reduction + scalar stage.
The vector-ALU forms are bit-identical to the serial order (LOOKAHEAD 1, OVERLAP 0), the matrix-core form agrees with it to
1e-12, through apply passes, runs in pieces, flushes, direct updates in between, a failing cut, observers; all within the
north-star tolerance of the oracle."""
import os

import numpy as np
import pytest

from test_gpu_resident import _cuts
from util import TOL, assert_state_close, dense_spd, set_default

pytestmark = pytest.mark.gpu


def _beta(b0, b1, i):
    return (b0[i], None if np.isnan(b1[i]) else b1[i])


# (OVERLAP, LOOKAHEAD, QUEUE_DEPTH); the first is the serial reference
MODES = [(0, 1, 0), (1, 1, 48), (0, 2, 0), (1, 3, 48), (0, 4, 0), (1, 12, 0), (0, 12, 48), (0, 16, 48), (1, 21, 48), (1, 32, 48)]
EXACT = 4                                                            # the first four: bit-identical to one another
RTOL = 1e-12                                                         # the matrix-core groups against them


def _same(a, b, exact):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    if exact:
        return np.array_equal(a, b)
    return np.max(np.abs(a - b)) <= RTOL * max(np.max(np.abs(a)), 1e-300)


def _drive(gpu, n, depth, mode, kinds, grads, b0, b1, pieces, direct=(), flush_after=(), q0=None):
    e = gpu.Ell.new_with_scalar(1.0, np.linspace(-1.0, 1.0, n)) if q0 is None else \
        gpu.Ell.new_with_matrix(1.0, q0, np.linspace(-1.0, 1.0, n))
    e.defer_depth = depth
    e.set_option(gpu.capi.OPT_OVERLAP, mode[0])
    e.set_option(gpu.capi.OPT_LOOKAHEAD, mode[1])
    e.set_option(gpu.capi.OPT_QUEUE_DEPTH, mode[2])
    e.queue_upload(kinds, grads, b0, b1)
    for a, c in pieces:
        if a in direct:      # a synchronous update of cut a between two runs (the run then starts at a + 1)
            assert int(e._update(int(kinds[a]), (grads[a], _beta(b0, b1, a)))) == 0
            a, c = a + 1, c - 1
        e.queue_run(a, c, fused=True)
        if a in flush_after:
            e.flush()
    st, ts = e.queue_results()
    return e, st, ts


@pytest.mark.parametrize("n,depth,dense", [(512, 8, False), (1024, 16, False), (1024, 24, False), (2050, 24, False),
                                           (4096, 24, False), (1024, 24, True), (2112, 24, True)])
def test_overlapped_runs_equal_the_serial_issue_order_to_the_bit(gpu, orc, n, depth, dense):
    """dense: from a dense SPD start matrix (new_with_matrix) -- the first group's products already multiply a full matrix
    (every tile of k_symm_mfma / k_symv carries data from the first pass on), not the identity."""
    set_default("SYMV_MIN_N", 512)
    set_default("RESIDENT", 0)
    k = 70
    kinds, grads, b0, b1 = _cuts(n, k, 31 * n + depth)
    q0 = dense_spd(n, 5 * n + depth) if dense else None
    pieces = [(0, 9), (9, 1), (10, 33), (43, 27)]
    st_ok = np.arange(k) != 43
    outs = []
    for mode in MODES:
        e, st, ts = _drive(gpu, n, depth, mode, kinds, grads, b0, b1, pieces, direct=(43,), flush_after=(10,), q0=q0)
        assert np.all(st[:43] == 0) and np.all(st[44:] == 0)
        outs.append((st, ts, e.xc(), e.kappa, e.mq, e))
    a = outs[0]
    for mi, (mode, b) in enumerate(zip(MODES, outs)):
        exact = mi < EXACT or n % 64 != 0      # (the matrix-core kernel needs n % 64 == 0: other sizes stay on the vector ALU)
        assert np.array_equal(a[0], b[0]) and _same(a[1][st_ok], b[1][st_ok], exact) and _same(a[2], b[2], exact), mode
        assert _same([a[3]], [b[3]], exact) and _same(a[4], b[4], exact), mode
    b = outs[-2]
    o = orc.OracleEll.new_with_scalar(1.0, np.linspace(-1.0, 1.0, n)) if q0 is None else \
        orc.OracleEll.new_with_matrix(1.0, q0, np.linspace(-1.0, 1.0, n))
    for i in range(k):
        assert o.update(int(kinds[i]), grads[i], b0[i], None if np.isnan(b1[i]) else b1[i]) == 0
        if i != 43:
            assert abs(b[1][i] - o.tsq) <= TOL * abs(o.tsq), i
    assert_state_close(b[5], o, what=f"overlapped n={n} depth={depth} dense={dense}")
    if dense:   # ... and the default mode (the last one: lookahead 32, 48 per apply pass) as well
        assert_state_close(outs[-1][5], o, what=f"lookahead 32 n={n} depth={depth} dense start")


@pytest.mark.parametrize("depth", [8, 24])
def test_failing_cut_halts_the_overlapped_run(gpu, orc, depth):
    """The GEMVs of the cuts after the failing one are already done or in flight when the scalar stage halts the queue:
    nothing of them is used, the state is the oracle's after the last successful cut, and the queue works again once the results are read."""
    set_default("SYMV_MIN_N", 512)
    set_default("RESIDENT", 0)
    n, k, bad = 1024, 40, 17
    kinds, grads, b0, b1 = _cuts(n, k, 77 + depth, fail_at=bad)
    res = []
    for mode in MODES:
        e, st, ts = _drive(gpu, n, depth, mode, kinds, grads, b0, b1, [(0, 30), (30, 10)])
        assert list(st[:bad]) == [0] * bad and int(st[bad]) == 1 and all(int(x) == 3 for x in st[bad + 1:])
        i = bad + 1
        assert int(e._update(int(kinds[i]), (grads[i], _beta(b0, b1, i)))) == 0
        e.queue_run(bad + 2, 10, fused=True)
        st2, ts2 = e.queue_results()
        assert np.all(st2[bad + 2:bad + 12] == 0)
        res.append((ts, ts2, e.xc(), e.kappa, e.mq, e))
    for mi, (mode, other) in enumerate(zip(MODES, res)):
        for x, y in zip(res[0][:5], other[:5]):
            assert _same(np.atleast_1d(x)[np.isfinite(np.atleast_1d(x))], np.atleast_1d(y)[np.isfinite(np.atleast_1d(y))],
                         mi < EXACT), mode
    o = orc.OracleEll.new_with_scalar(1.0, np.linspace(-1.0, 1.0, n))
    for i in list(range(bad)) + list(range(bad + 1, bad + 12)):
        assert o.update(int(kinds[i]), grads[i], b0[i], None if np.isnan(b1[i]) else b1[i]) == 0
    assert_state_close(res[1][5], o, what="after the halt")


def test_overlapped_run_at_the_default_size_and_depth(gpu):
    """n = 8192: the smallest size that takes the lower-triangle schedule (depth 24, groups of up to 32 on the matrix cores) by
    itself; a clone taken between two runs continues serially and must stay equal to 1e-12."""
    from ellalgo_rs_amd import synth
    n, k = 8192, 60
    kinds, grads, b0, b1 = synth.parallel_cuts(n, k)
    e = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    assert e.defer_depth == 24 and e.get_option(gpu.capi.OPT_OVERLAP) == 1 and e.get_option(gpu.capi.OPT_LOOKAHEAD) == 32
    e.profile_enable(True)
    e.queue_upload(kinds, grads, b0, b1)
    e.queue_run(0, 31, fused=True)
    c = e.clone()
    c.set_option(gpu.capi.OPT_OVERLAP, 0)
    c.set_option(gpu.capi.OPT_LOOKAHEAD, 1)
    c.queue_upload(kinds, grads, b0, b1)
    e.queue_run(31, 29, fused=True)
    c.queue_run(31, 29, fused=True)
    st_e, ts_e = e.queue_results()
    st_c, ts_c = c.queue_results()
    assert np.all(st_e == 0) and np.all(st_c[31:] == 0)
    assert _same(ts_e[31:], ts_c[31:], False) and _same(e.xc(), c.xc(), False) and _same([e.kappa], [c.kappa], False)
    assert _same(e.mq, c.mq, False)
    prof = e.profile_read()
    # one pass over Q and one batched reduction per group of up to 32 cuts (a run's end closes a group):
    # (31) + (29); both runs end with more than 24 recorded and apply them before they return
    assert prof["symv_reduce"][1] == 2 and prof["symv"][1] == 2 and prof["apply"][1] == 2


# Seeds the 400-seed soak of round 3 failed on with the library as it was before 957140e (products issued ahead on the second
# stream were not ordered after an apply pass of the single-cut path): found again in round 4 by running that soak against
# the old library (tools/experiments/README_soak_r03.md), pinned here so the default suite always walks them.
PINNED_SEEDS = [39, 99, 150, 191, 203, 332, 336, 369]
_NSEEDS = int(os.environ.get("ELLHIP_FUZZ_OPTION_SEEDS", "24"))   # (soak: ELLHIP_FUZZ_OPTION_SEEDS=400)
_SEEDS = list(range(_NSEEDS)) + [x for x in PINNED_SEEDS if x >= _NSEEDS]


def _option_walk(gpu, seed, serial):
    """One seeded walk (see test_random_option_mixes_against_the_oracle).  serial: every ELLHIP_OPT_OVERLAP = 1 of the walk
    becomes 2 -- the same kernels in the same order of issue, all on the handle's own stream."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([512, 576, 640, 1000, 1024, 1090, 2112, 4096 if seed % 6 == 0 else 1536]))
    depth = int(rng.choice([8, 16, 24, 24]))
    k = int(rng.integers(50, 110))
    bad = int(rng.integers(10, k)) if rng.random() < 0.4 else None
    kinds, grads, b0, b1 = _cuts(n, k, 17 * seed + n, fail_at=bad)
    e = gpu.Ell.new_with_scalar(1.0, np.linspace(-1.0, 1.0, n))
    e.defer_depth = depth

    def reroll():
        e.set_option(gpu.capi.OPT_LOOKAHEAD, int(rng.choice([1, 2, 3, 4, 7, 12, 16, 21, 32])))
        e.set_option(gpu.capi.OPT_QUEUE_DEPTH, int(rng.choice([0, 48])))
        ov = int(rng.integers(0, 2))
        e.set_option(gpu.capi.OPT_OVERLAP, 2 if (serial and ov) else ov)

    reroll()
    e.queue_upload(kinds, grads, b0, b1)
    stop = bad if bad is not None else k     # cuts [0, stop) succeed
    direct = set()
    pos = 0
    while pos < k:
        step = int(rng.integers(1, min(60, k - pos) + 1))
        act = rng.random()
        if act < 0.15 and pos < stop - 1 and (bad is None or pos < bad):
            # a synchronous update of cut `pos` (only while the queue has not halted: a halted queue refuses nothing, but the
            # oracle's sequence has no cut behind the failing one)
            assert int(e._update(int(kinds[pos]), (grads[pos], _beta(b0, b1, pos)))) == 0
            direct.add(pos)            # never ran in the queue
            pos += 1
            continue
        e.queue_run(pos, step, fused=True)
        pos += step
        r = rng.random()
        if r < 0.2:
            e.flush()
        elif r < 0.3:
            reroll()
        elif r < 0.4 and (bad is None or pos <= bad):
            assert abs(e.kappa - 0.0) >= 0.0 and e.xc().shape == (n,)   # observers in the middle of the sequence
    st, ts = e.queue_results()
    return dict(n=n, depth=depth, k=k, bad=bad, cuts=(kinds, grads, b0, b1), direct=direct, e=e, st=st, ts=ts)


@pytest.mark.parametrize("seed", _SEEDS)
def test_random_option_mixes_against_the_oracle(gpu, orc, seed):
    """Seeded walks over what the queue run can be asked to do: size (multiples of 64 and not, both segment widths), depth,
    LOOKAHEAD / QUEUE_DEPTH / OVERLAP, the run cut into random pieces with direct updates, flushes, option switches and
    observers in between, a failing cut at a random place; every cut's status and tsq and the final state against the
    oracle's plain sequence of updates (north-star tolerance).  Every walk runs twice: with the second stream, and with the
    same kernels issued in the same order on ONE stream (ELLHIP_OPT_OVERLAP = 2); the two must agree to the bit -- nothing in
    the data path depends on timing, so any difference is a missing cross-stream ordering."""
    set_default("SYMV_MIN_N", 512)
    set_default("RESIDENT", 0)
    w = _option_walk(gpu, seed, serial=False)
    ws = _option_walk(gpu, seed, serial=True)
    n, k, bad, e = w["n"], w["k"], w["bad"], w["e"]
    assert np.array_equal(w["st"], ws["st"]) and np.array_equal(w["ts"], ws["ts"], equal_nan=True), f"seed {seed}: queue results"
    assert np.array_equal(e.xc(), ws["e"].xc()) and e.kappa == ws["e"].kappa, f"seed {seed}: xc / kappa overlapped vs serial"
    assert np.array_equal(e.mq, ws["e"].mq), f"seed {seed}: Q overlapped vs one stream"
    kinds, grads, b0, b1 = w["cuts"]
    o = orc.OracleEll.new_with_scalar(1.0, np.linspace(-1.0, 1.0, n))
    st, ts = w["st"], w["ts"]
    halted = False
    for i in range(k):
        if halted:
            assert int(st[i]) == 3, (i, int(st[i]), bad)
            continue
        so = o.update(int(kinds[i]), grads[i], b0[i], None if np.isnan(b1[i]) else b1[i])
        halted = so != 0
        if i in w["direct"]:
            continue
        assert int(st[i]) == so, (i, int(st[i]), so, bad)
        assert abs(ts[i] - o.tsq) <= TOL * abs(o.tsq), (i, bad)
    assert_state_close(e, o, what=f"seed {seed}: n={n} depth={w['depth']} bad={bad}")
