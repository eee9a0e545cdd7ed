"""GPU rehearsal of the row-partitioned schedule with the real HIP engine.

Only one MI355X is available to the tests, so:
  * two ranks share cuda:0 and exchange their gt slices through a gloo (CPU) all-gather -- this runs
    HipShardEngine, the torch-stream plumbing and the orchestration exactly as in production, with
    only the transport swapped;
  * a single rank runs the production exchange itself (RCCL, in-place all_gather_into_tensor) in a
    world of one, which checks the call is accepted and ordered correctly on the engine's stream.
Both must reproduce the unsharded engine bit for bit."""
import os
import socket
import sys
import traceback

import numpy as np
import pytest

import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, backend, n, k, fused, defer, errq, symmetric=False):
    try:
        for p in (ROOT, HERE):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        import ellalgo_rs_amd as pkg
        from ellalgo_rs_amd import synth
        from ellalgo_rs_amd.sharded import ShardedEll

        def bounce(gt, row0, nrows):  # gloo transport for two ranks on one card
            torch.cuda.current_stream().synchronize()
            mine = gt[row0:row0 + nrows].cpu()
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            gt.copy_(torch.cat(parts).to(gt.device))

        def bounce_sum(gt, row0, nrows):  # symmetric schedule: all-reduce of the partial vectors, via gloo
            torch.cuda.current_stream().synchronize()
            t = gt.cpu()
            dist.all_reduce(t)
            gt.copy_(t.to(gt.device))

        kinds, grads, b0, b1 = synth.parallel_cuts(n, k)
        # row shards stream; their unsharded reference must too for a comparison of bits (the resident queue run of an
        # unsharded handle sums Q g in its own shape: test_gpu_resident.py)
        pkg.capi.set_default_option(pkg.capi.OPT_RESIDENT, 0)
        ref = pkg.Ell.new_with_scalar(1.0, np.zeros(n), device=0)
        ex = None if backend == "nccl" else (bounce_sum if symmetric else bounce)
        sh = ShardedEll.new_with_scalar(1.0, np.zeros(n), device=0, exchange=ex, symmetric=symmetric)
        if symmetric:
            from ellalgo_rs_amd.sharded import partition_symmetric
            assert (sh.row0, sh.nrows) == partition_symmetric(n, world, rank)
        if defer != 1:
            ref.defer_depth = defer
            if not symmetric:   # (the symmetric constructor has selected depth 8 already)
                sh.set_defer_depth(defer)
        half = k // 2
        # depth 8: the unsharded reference takes the lower-triangle GEMV, a shard the full-row one: same
        # vector to rounding.  depth 1: identical kernels, identical bits.
        exact = defer == 1

        def same(a, b):
            return np.array_equal(a, b) if exact else np.max(np.abs(np.asarray(a) - np.asarray(b))) <= 1e-12 * np.max(np.abs(b))

        for i in range(half):  # direct, synchronous updates
            cut = (grads[i], (b0[i], b1[i]))
            assert int(sh._update(int(kinds[i]), cut)) == int(ref._update(int(kinds[i]), cut)) == 0
            assert same([sh.tsq()], [ref.tsq()])
        # then the device-resident queue
        sh.queue_upload(kinds[half:], grads[half:], b0[half:], b1[half:])
        ref.queue_upload(kinds[half:], grads[half:], b0[half:], b1[half:])
        sh.queue_run(0, 2, fused=fused)
        if defer != 1:
            # as bench.py does between its regions: apply the recorded updates while (pipelined schedule) cut 2 is
            # primed -- the shard drops that GEMV (it belonged to the old base) and the next run primes + exchanges again
            sh.flush()
            assert sh._primed_index == -1
        sh.queue_run(2, k - half - 2, fused=fused)
        ref.queue_run(0, k - half)
        st_s, ts_s = sh.queue_results()
        st_r, ts_r = ref.queue_results()
        assert np.array_equal(st_s, st_r) and same(ts_s, ts_r) and np.all(st_r == 0)
        mine, want = sh.mq_rows, ref.mq[sh.row0:sh.row0 + sh.nrows]
        if symmetric:   # rows are current up to their diagonal only; the mirrored half lives on other ranks
            keep = np.arange(n)[None, :] <= (sh.row0 + np.arange(sh.nrows))[:, None]
            mine, want = np.where(keep, mine, 0.0), np.where(keep, want, 0.0)
        assert same(mine, want), "Q rows differ from unsharded engine"
        assert same(sh.xc(), ref.xc()) and same([sh.kappa], [ref.kappa])
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        errq.put((rank, traceback.format_exc()))
        raise


def _run(world, backend, n, k, fused=False, defer=1, symmetric=False):
    ctx = mp.get_context("spawn")
    errq = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, n, k, fused, defer, errq, symmetric))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    errs = []
    while not errq.empty():
        errs.append(errq.get())
    for p in procs:
        if p.is_alive():
            p.terminate()
            errs.append((-1, "worker timed out"))
    assert not errs, "\n".join(f"[rank {r}] {t}" for r, t in errs)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]


def test_two_ranks_one_gpu_gloo_transport(gpu):
    _run(2, "gloo", 512, 12)


def test_two_ranks_one_gpu_pipelined_schedule(gpu):
    _run(2, "gloo", 512, 12, fused=True)


def test_two_ranks_one_gpu_deferred_shrink(gpu):
    _run(2, "gloo", 512, 40, fused=True, defer=8)


def test_one_rank_rccl_deferred_shrink(gpu):
    _run(1, "nccl", 1024, 40, fused=True, defer=8)


def test_one_rank_rccl_in_place_allgather(gpu):
    _run(1, "nccl", 1024, 12)


def test_one_rank_rccl_pipelined_schedule(gpu):
    _run(1, "nccl", 1024, 12, fused=True)


@pytest.mark.parametrize("world,n,fused", [(2, 512, True), (2, 1024, False), (3, 1536, True)])
def test_symmetric_shards_share_one_gpu(gpu, world, n, fused):
    """ellhip_set_shard_symmetric: sqrt row partition, partial lower-trapezoid GEMVs added by an all-reduce (gloo
    transport between the ranks sharing the card), lower-trapezoid apply passes; against the unsharded engine."""
    _run(world, "gloo", n, 40, fused=fused, defer=8, symmetric=True)


def test_symmetric_shard_one_rank_rccl_all_reduce(gpu):
    _run(1, "nccl", 1024, 40, fused=True, defer=8, symmetric=True)


def test_symmetric_shard_refuses_other_schedules(gpu):
    import ctypes as C
    lib = gpu.capi.load()
    n = 256
    h = C.c_void_p()
    gpu.capi.check(lib.ellhip_create_shard(C.byref(h), n, 64, 128, 1.0, None, None, None, -1))
    gpu.capi.check(lib.ellhip_set_shard_symmetric(h, 1))
    g = np.ones(n)
    # depth 1: refused loudly (the all-reduce data flow needs the partial symmetric GEMV)
    assert lib.ellhip_update_begin(h, 0, g.ctypes.data_as(C.c_void_p), 0.01, 0, 0.0) == gpu.capi.E_STATE
    assert lib.ellhip_set_no_defer_trick(h, 1) == gpu.capi.E_STATE
    lib.ellhip_destroy(h)
    gpu.capi.check(lib.ellhip_create_shard(C.byref(h), n, 32, 128, 1.0, None, None, None, -1))
    assert lib.ellhip_set_shard_symmetric(h, 1) == gpu.capi.E_INVALID   # boundaries must be multiples of 64
    lib.ellhip_destroy(h)
    e = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
    assert lib.ellhip_set_shard_symmetric(e._h, 1) == gpu.capi.E_INVALID  # not a shard
