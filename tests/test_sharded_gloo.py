"""world_size-2 (and 4) CPU tests of the row-partitioned schedule over the gloo backend.

The product orchestration (ellalgo_rs_amd.sharded.ShardedEll: partition, in-place all-gather of the
gt slices, redundant scalar stage, local rank-1, queue loop) runs unchanged; only the per-rank engine
is the oracle-backed test double, because there is no GPU here.  Every rank must end up with its row
block of exactly the matrix the single-process oracle produces (bit-identical)."""
import os
import socket
import sys
import traceback

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, k, mode, errq):
    try:
        for p in (ROOT, HERE):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import ellalgo_rs_amd as pkg
        from ellalgo_rs_amd.sharded import ShardedEll, partition
        from oracle import oracle
        from shard_engines import OracleShardEngine
        from util import mixed_cut

        rng = np.random.default_rng(42)  # same stream on every rank
        xc0 = np.linspace(-1.0, 1.0, n)
        ref = oracle.OracleEll.new_with_scalar(2.0, xc0)
        symmetric = mode.startswith("sym_")
        mode = mode[4:] if symmetric else mode
        sh = ShardedEll.new_with_scalar(2.0, xc0, engine_factory=OracleShardEngine, symmetric=symmetric)
        if symmetric:
            from ellalgo_rs_amd.sharded import partition_symmetric
            row0, nrows = partition_symmetric(n, world, rank)
        else:
            row0, nrows = partition(n, world, rank)
        assert (sh.row0, sh.nrows) == (row0, nrows)
        cuts = []
        for i in range(k):
            g = rng.standard_normal(n)
            g /= np.linalg.norm(g)
            tau = float(np.sqrt(max(ref.kappa * (g @ (ref.mq @ g)), 0.0)))
            kind, b0, b1 = mixed_cut(i, g, tau, rng)
            cuts.append((kind, g, b0, b1))
            so = ref.update(kind, g, b0, b1)
            if mode == "direct":
                ss = sh._update(kind, (g, (b0, b1)))
                assert int(ss) == so, (i, int(ss), so)
                if symmetric:   # partial sums added across ranks: same value, another association
                    assert abs(sh.tsq() - ref.tsq) <= 1e-12 * abs(ref.tsq)
                else:
                    assert sh.tsq() == ref.tsq
        if mode in ("queue", "queue_fused", "queue_fused_observed"):
            # the queue halts at the first failure, so replay only the successful prefix rule: build a
            # fresh reference that stops like the drivers do
            ref = oracle.OracleEll.new_with_scalar(2.0, xc0)
            kinds = np.array([c[0] for c in cuts], dtype=np.int32)
            grads = np.array([c[1] for c in cuts])
            b0s = np.array([c[2] for c in cuts])
            b1s = np.array([np.nan if c[3] is None else c[3] for c in cuts])
            sh.queue_upload(kinds, grads, b0s, b1s)
            if mode == "queue":
                sh.queue_run(0, k)
            elif mode == "queue_fused_observed":
                # as bench.py drives a shard on a recorded schedule: flush / look at the rows / switch the depth between
                # pipelined runs; each of those drops the shard's primed GEMV (here: poisoned with NaN), and every rank
                # has to prime -- and exchange -- again before the next cut
                sh.engine.drop_on_flush = True
                sh.queue_run(0, 3, fused=True)
                sh.flush()
                assert sh._primed_index == -1
                sh.queue_run(3, 4, fused=True)
                _ = sh.mq_rows
                assert sh._primed_index == -1
                sh.queue_run(7, 2, fused=True)
                sh.set_defer_depth(8)
                sh.queue_run(9, k - 9, fused=True)
                sh.engine.drop_on_flush = False
            else:   # pipelined schedule, split in two calls like the benchmark (warm-up, then timed)
                sh.queue_run(0, 3, fused=True)
                sh.queue_run(3, k - 3, fused=True)
            st, ts = sh.queue_results()
            halted = False
            for i, (kind, g, b0, b1) in enumerate(cuts):
                if halted:
                    assert st[i] == 3
                    continue
                so = ref.update(kind, g, b0, b1)
                assert st[i] == so
                assert ts[i] == ref.tsq or (symmetric and abs(ts[i] - ref.tsq) <= 1e-12 * abs(ref.tsq))
                halted = so != 0
        if symmetric:
            assert np.allclose(sh.mq_rows, ref.mq[row0:row0 + nrows], rtol=1e-11, atol=1e-14), "Q rows differ"
            assert np.allclose(sh.xc(), ref.xc, rtol=1e-11, atol=1e-14), "xc differs"
            assert abs(sh.kappa - ref.kappa) <= 1e-12 * ref.kappa
        else:   # every rank: its rows of Q, the full xc and kappa, bit for bit
            assert np.array_equal(sh.mq_rows, ref.mq[row0:row0 + nrows]), "Q rows differ"
            assert np.array_equal(sh.xc(), ref.xc), "xc differs"
            assert sh.kappa == ref.kappa
        # and the ranks agree with each other
        xs = [torch.zeros(n, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(xs, torch.from_numpy(sh.xc()))
        assert all(torch.equal(xs[0], x) for x in xs)
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        errq.put((rank, traceback.format_exc()))
        raise


def _run(world, n, k, mode):
    ctx = mp.get_context("spawn")
    errq = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, k, mode, errq)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    errs = []
    while not errq.empty():
        errs.append(errq.get())
    for p in procs:
        if p.is_alive():
            p.terminate()
            errs.append((-1, "worker timed out"))
    assert not errs, "\n".join(f"[rank {r}] {t}" for r, t in errs)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]


@pytest.mark.parametrize("world,n", [(2, 64), (4, 96)])
def test_sharded_direct_updates_bit_identical(world, n):
    _run(world, n, 24, "direct")


def test_sharded_queue_bit_identical_and_halts():
    _run(2, 48, 16, "queue")


def test_sharded_pipelined_queue_bit_identical_and_halts():
    _run(2, 48, 16, "queue_fused")


@pytest.mark.parametrize("mode,n", [("queue_fused_observed", 64), ("sym_queue_fused_observed", 192)])
def test_sharded_pipelined_queue_with_observers_between_runs(mode, n):
    _run(2, n, 20, mode)


@pytest.mark.parametrize("world,n,mode", [(2, 128, "sym_direct"), (3, 320, "sym_direct"), (2, 192, "sym_queue_fused"),
                                          (4, 512, "sym_queue")])
def test_symmetric_sharding_all_reduce_schedule(world, n, mode):
    """Row blocks of equal trapezoid area, partial symmetric GEMVs, ONE all-reduce per update (gloo)."""
    _run(world, n, 16, mode)


def test_symmetric_partition_rules():
    sys.path.insert(0, ROOT)
    from ellalgo_rs_amd.sharded import partition_symmetric
    for n, world in ((16384, 2), (16384, 8), (32768, 8), (512, 3), (128, 2)):
        parts = [partition_symmetric(n, world, r) for r in range(world)]
        assert parts[0][0] == 0 and sum(p[1] for p in parts) == n
        assert all(p[0] % 64 == 0 and p[1] % 64 == 0 and p[1] > 0 for p in parts)
        assert all(parts[r + 1][0] == parts[r][0] + parts[r][1] for r in range(world - 1))
        if n >= 16384:
            areas = [((r0 + nr) ** 2 - r0 ** 2) / 2 for r0, nr in parts]
            assert max(areas) <= 1.05 * n * n / 2 / world
    with pytest.raises(ValueError):
        partition_symmetric(100, 2, 0)


def test_partition_rules():
    sys.path.insert(0, ROOT)
    from ellalgo_rs_amd.sharded import partition
    assert partition(32768, 8, 3) == (3 * 4096, 4096)
    assert [partition(12, 3, r) for r in range(3)] == [(0, 4), (4, 4), (8, 4)]
    with pytest.raises(ValueError):
        partition(10, 4, 0)
    with pytest.raises(ValueError):
        partition(8, 2, 2)
