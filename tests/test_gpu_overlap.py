"""GPU: pipelined queue runs on the lower-triangle schedule that use what a QUEUE knows -- the next gradients.
The GEMV y = Q_base g of a queued cut reads a matrix the cuts before it do not change until the next apply pass (recorded
schedule: src/ell.rs:117-128 is deferred), so
  * ELLHIP_OPT_LOOKAHEAD = L (default 16; csrc/ellhip_capi.hip queue_run_multi): the products of L consecutive queued cuts
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
from util import TOL, assert_state_close, set_default

pytestmark = pytest.mark.gpu


def _beta(b0, b1, i):
    return (b0[i], None if np.isnan(b1[i]) else b1[i])


# (OVERLAP, LOOKAHEAD, QUEUE_DEPTH); the first is the serial reference
MODES = [(0, 1, 0), (1, 1, 48), (0, 2, 0), (1, 3, 48), (0, 4, 0), (1, 12, 0), (0, 12, 48), (0, 16, 48)]
EXACT = 4                                                            # the first four: bit-identical to one another
RTOL = 1e-12                                                         # the matrix-core groups against them


def _same(a, b, exact):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    if exact:
        return np.array_equal(a, b)
    return np.max(np.abs(a - b)) <= RTOL * max(np.max(np.abs(a)), 1e-300)


def _drive(gpu, n, depth, mode, kinds, grads, b0, b1, pieces, direct=(), flush_after=()):
    e = gpu.Ell.new_with_scalar(1.0, np.linspace(-1.0, 1.0, n))
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


@pytest.mark.parametrize("n,depth", [(512, 8), (1024, 16), (1024, 24), (2050, 24), (4096, 24)])
def test_overlapped_runs_equal_the_serial_issue_order_to_the_bit(gpu, orc, n, depth):
    set_default("SYMV_MIN_N", 512)
    set_default("RESIDENT", 0)
    k = 70
    kinds, grads, b0, b1 = _cuts(n, k, 31 * n + depth)
    pieces = [(0, 9), (9, 1), (10, 33), (43, 27)]
    st_ok = np.arange(k) != 43
    outs = []
    for mode in MODES:
        e, st, ts = _drive(gpu, n, depth, mode, kinds, grads, b0, b1, pieces, direct=(43,), flush_after=(10,))
        assert np.all(st[:43] == 0) and np.all(st[44:] == 0)
        outs.append((st, ts, e.xc(), e.kappa, e.mq, e))
    a = outs[0]
    for mi, (mode, b) in enumerate(zip(MODES, outs)):
        exact = mi < EXACT or n % 64 != 0      # (the matrix-core kernel needs n % 64 == 0: other sizes stay on the vector ALU)
        assert np.array_equal(a[0], b[0]) and _same(a[1][st_ok], b[1][st_ok], exact) and _same(a[2], b[2], exact), mode
        assert _same([a[3]], [b[3]], exact) and _same(a[4], b[4], exact), mode
    b = outs[-2]
    o = orc.OracleEll.new_with_scalar(1.0, np.linspace(-1.0, 1.0, n))
    for i in range(k):
        assert o.update(int(kinds[i]), grads[i], b0[i], None if np.isnan(b1[i]) else b1[i]) == 0
        if i != 43:
            assert abs(b[1][i] - o.tsq) <= TOL * abs(o.tsq), i
    assert_state_close(b[5], o, what=f"overlapped n={n} depth={depth}")


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
    """n = 8192: the smallest size that takes the lower-triangle schedule (depth 24, groups of 16 on the matrix cores) by
    itself; a clone taken between two runs continues serially and must stay equal to 1e-12."""
    from ellalgo_rs_amd import synth
    n, k = 8192, 60
    kinds, grads, b0, b1 = synth.parallel_cuts(n, k)
    e = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    assert e.defer_depth == 24 and e.get_option(gpu.capi.OPT_OVERLAP) == 1 and e.get_option(gpu.capi.OPT_LOOKAHEAD) == 16
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
    # one pass over Q and one batched reduction per group of up to sixteen cuts (a run's end closes a group early):
    # (16, 15) + (16, 13); both runs end with more than 24 recorded and apply them before they return
    assert prof["symv_reduce"][1] == 4 and prof["symv"][1] == 4 and prof["apply"][1] == 2


@pytest.mark.parametrize("seed", range(int(os.environ.get("ELLHIP_FUZZ_OPTION_SEEDS", "24"))))   # (soak: ELLHIP_FUZZ_OPTION_SEEDS=400)
def test_random_option_mixes_against_the_oracle(gpu, orc, seed):
    """Seeded walks over what the queue run can be asked to do: size (multiples of 64 and not, both segment widths), depth,
    LOOKAHEAD / QUEUE_DEPTH / OVERLAP, the run cut into random pieces with direct updates, flushes, option switches and
    observers in between, a failing cut at a random place; every cut's status and tsq and the final state against the
    oracle's plain sequence of updates (north-star tolerance)."""
    rng = np.random.default_rng(1000 + seed)
    set_default("SYMV_MIN_N", 512)
    set_default("RESIDENT", 0)
    n = int(rng.choice([512, 576, 640, 1000, 1024, 1090, 2112, 4096 if seed % 6 == 0 else 1536]))
    depth = int(rng.choice([8, 16, 24, 24]))
    k = int(rng.integers(50, 110))
    bad = int(rng.integers(10, k)) if rng.random() < 0.4 else None
    kinds, grads, b0, b1 = _cuts(n, k, 17 * seed + n, fail_at=bad)
    e = gpu.Ell.new_with_scalar(1.0, np.linspace(-1.0, 1.0, n))
    e.defer_depth = depth

    def reroll():
        e.set_option(gpu.capi.OPT_LOOKAHEAD, int(rng.choice([1, 2, 3, 4, 7, 12, 16])))
        e.set_option(gpu.capi.OPT_QUEUE_DEPTH, int(rng.choice([0, 48])))
        e.set_option(gpu.capi.OPT_OVERLAP, int(rng.integers(0, 2)))

    reroll()
    e.queue_upload(kinds, grads, b0, b1)
    o = orc.OracleEll.new_with_scalar(1.0, np.linspace(-1.0, 1.0, n))
    want_st, want_ts, halted = [], [], False
    for i in range(k):
        if halted:
            want_st.append(3)
            want_ts.append(None)
            continue
        so = o.update(int(kinds[i]), grads[i], b0[i], None if np.isnan(b1[i]) else b1[i])
        want_st.append(so)
        want_ts.append(o.tsq)
        halted = so != 0
    stop = bad if bad is not None else k     # cuts [0, stop) succeed
    pos = 0
    while pos < k:
        step = int(rng.integers(1, min(60, k - pos) + 1))
        act = rng.random()
        if act < 0.15 and pos < stop - 1 and (bad is None or pos < bad):
            # a synchronous update of cut `pos` (only while the queue has not halted: a halted queue refuses nothing, but the
            # oracle's sequence has no cut behind the failing one)
            assert int(e._update(int(kinds[pos]), (grads[pos], _beta(b0, b1, pos)))) == 0
            want_st[pos] = -1          # never ran in the queue
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
    for i in range(k):
        if want_st[i] == -1:
            continue
        assert int(st[i]) == want_st[i], (i, int(st[i]), want_st[i], bad)
        if want_ts[i] is not None:
            assert abs(ts[i] - want_ts[i]) <= TOL * abs(want_ts[i]), (i, bad)
    assert_state_close(e, o, what=f"seed {seed}: n={n} depth={depth} bad={bad}")
