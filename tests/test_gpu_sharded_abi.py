"""GPU: the row-partitioned Ell behind the C ABI (include/ellhip_sharded.h), where libellhip.so itself issues the
collective through RCCL.  One MI355X is available to the tests, and RCCL refuses two ranks on one device, so what runs
here is a group of ONE rank whose communicator the library creates from its own unique id: ncclCommInitRank,
ncclAllGather / ncclAllReduce on the handle's stream, ordering against the passes before and after, and the whole
orchestration (direct updates, the queue two-pass and pipelined, observers) -- against the unsharded engine.
The arithmetic of several shards side by side is covered by tests/test_gpu_sharded.py (two and three ranks sharing the
card over gloo) and tests/cpp/sharded_runner.cpp (in-process shards through the C++ class)."""
import numpy as np
import pytest

from util import TOL, assert_state_close, run_mixed, set_default

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_id(gpu):
    from ellalgo_rs_amd import sharded_abi
    return sharded_abi.unique_id()


def _fresh_id():
    from ellalgo_rs_amd import sharded_abi
    return sharded_abi.unique_id()


class _AsSpace:
    """adapter: ShardedEllAbi of one rank looks like an unsharded space to the parity helpers"""
    def __init__(self, s):
        self.s, self.n = s, s.n

    def _update(self, kind, cut):
        return self.s._update(kind, cut)

    def xc(self):
        return self.s.xc()

    def tsq(self):
        return self.s.tsq()

    @property
    def kappa(self):
        return self.s.kappa

    @property
    def mq(self):
        return self.s.mq_rows


@pytest.mark.parametrize("n", [64, 192, 1000, 2048])
def test_one_rank_equal_blocks_matches_the_oracle(gpu, orc, n):
    s = gpu.ShardedEllAbi.new_with_scalar(2.0, np.linspace(-1, 1, n), nccl_id=_fresh_id())
    o = orc.OracleEll.new_with_scalar(2.0, np.linspace(-1, 1, n))
    assert run_mixed(_AsSpace(s), o, 24, seed=40 + n, check_every=8) >= 12
    assert_state_close(_AsSpace(s), o, what=f"n={n}")


@pytest.mark.parametrize("depth", [1, 8])
def test_one_rank_equal_blocks_is_bit_identical_to_the_unsharded_engine(gpu, depth, monkeypatch):
    set_default("RESIDENT", 0)   # row shards stream; their unsharded reference must too for a comparison of bits
    set_default("AUTO_DEFER", 0)
    set_default("SYMV", 0)     # an equal-block shard runs full-row GEMVs: so must the reference here
    from ellalgo_rs_amd import synth
    n, k = 1536, 20
    kinds, grads, b0, b1 = synth.parallel_cuts(n, k)
    ref = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    ref.defer_depth = depth
    s = gpu.ShardedEllAbi.new_with_scalar(1.0, np.zeros(n), nccl_id=_fresh_id(), defer_depth=depth)
    for i in range(k // 2):
        cut = (grads[i], (b0[i], b1[i]))
        assert int(ref._update(int(kinds[i]), cut)) == int(s._update(int(kinds[i]), cut)) == 0
        assert ref.tsq() == s.tsq() and ref.kappa == s.kappa
    assert np.array_equal(ref.xc(), s.xc()) and np.array_equal(ref.mq, s.mq_rows)
    # the rest through the queue, pipelined
    for e in (ref, s):
        e.queue_upload(kinds, grads, b0, b1)
        e.queue_run(k // 2, k - k // 2, fused=True)
    (st1, ts1), (st2, ts2) = ref.queue_results(), s.queue_results()
    assert np.array_equal(st1[k // 2:], st2[k // 2:]) and np.all(st2[k // 2:] == 0)
    assert np.array_equal(ts1[k // 2:], ts2[k // 2:])
    assert np.array_equal(ref.xc(), s.xc()) and ref.kappa == s.kappa and np.array_equal(ref.mq, s.mq_rows)


@pytest.mark.parametrize("depth", [8, 16, 24])
def test_one_rank_symmetric_shard_all_reduce(gpu, orc, depth):
    """Symmetric partition: lower-triangle GEMV on the local trapezoid, ncclAllReduce of the partial vector, lower-
    trapezoid apply passes; with one rank the trapezoid is the whole triangle."""
    from ellalgo_rs_amd import synth
    n, k = 2048, 40
    kinds, grads, b0, b1 = synth.parallel_cuts(n, k)
    s = gpu.ShardedEllAbi.new_with_scalar(1.0, np.zeros(n), nccl_id=_fresh_id(), symmetric=True, defer_depth=depth)
    o = orc.OracleEll.new_with_scalar(1.0, np.zeros(n))
    s.queue_upload(kinds, grads, b0, b1)
    s.queue_run(0, 10, fused=False)
    s.queue_run(10, 17, fused=True)
    s.flush()                              # an observer between two pipelined runs: the prime is dropped and redone
    s.queue_run(27, k - 27, fused=True)
    st, ts = s.queue_results()
    want = []
    for i in range(k):
        assert o.update(int(kinds[i]), grads[i], b0[i], b1[i]) == 0
        want.append(o.tsq)
    assert np.all(st == 0) and np.max(np.abs(ts - np.array(want)) / np.abs(want)) <= TOL
    assert abs(s.kappa - o.kappa) <= TOL * abs(o.kappa)
    assert np.max(np.abs(s.xc() - np.array(o.xc))) <= TOL * np.max(np.abs(o.xc))
    q = s.mq_rows                          # rows are current up to their diagonal
    assert np.max(np.abs(np.tril(q) - np.tril(o.mq))) <= TOL * np.max(np.abs(o.mq))


def test_no_communicator_needed_for_one_rank(gpu, orc):
    n = 300
    s = gpu.ShardedEllAbi.new_with_scalar(1.0, np.zeros(n))      # nranks = 1, no id: no collective is issued
    o = orc.OracleEll.new_with_scalar(1.0, np.zeros(n))
    run_mixed(_AsSpace(s), o, 16, seed=3, check_every=4)


def test_cpp_host_drives_row_blocks_without_python(gpu):
    """tests/cpp/sharded_runner.cpp: EllShardGroup (2 and 3 row blocks on this card, in-process exchange), the
    reference driver on it, and ShardedEllHip over RCCL -- each bit-identical to the unsharded EllHip."""
    from cpp_build import build_runner, run_json_lines
    out = run_json_lines(build_runner("sharded_runner.cpp", "hip"))
    assert set(out) == {"group2", "group3", "driver", "rccl"}
    for name, d in out.items():
        assert d["ok"] is True, (name, d)
    assert out["group2"]["successes"] >= 20 and out["driver"]["niter"] > 5
