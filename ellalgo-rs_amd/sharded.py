"""Row-block partitioned Ell over the GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).  Rank r owns rows
[r*n/P, (r+1)*n/P) of Q; every rank keeps full copies of g, gt = Q*g, xc and the scalars.  One
update is

    phase 1  local GEMV                gt[R_r] = Q[R_r, :] * g            (engine.begin)
    exchange in-place all-gather of gt (n/P doubles per rank)             (the ONLY collective)
    phase 2  redundant scalar stage (omega, tsq, EllCalc, xc, kappa: identical bits on every rank,
             fixed reduction shapes) + rank-1 update of the local rows    (engine.end)

The per-rank engine and the exchange primitive are injected, so the orchestration below is the same
code over the HIP engine + RCCL on GPUs and in the world_size-2 gloo tests on CPU (where the tests
supply an oracle-backed engine; this module itself never imports the oracle and its default engine
is the HIP library, which fails loudly without a GPU).
"""
from __future__ import annotations

import contextlib
import ctypes as C

import numpy as np

from . import capi
from .ell import CutStatus, _f64, _p, _split


def partition(n: int, world: int, rank: int):
    """Contiguous equal row blocks; n must divide evenly so the all-gather is uniform."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank {rank} of {world}")
    if n % world:
        raise ValueError(f"n={n} is not divisible by world size {world}")
    nrows = n // world
    return rank * nrows, nrows


def partition_symmetric(n: int, world: int, rank: int, align: int = 64):
    """Row blocks with equal LOWER-TRAPEZOID area: boundaries at n*sqrt(r/world), rounded to `align` rows (the
    strip height of the lower-triangle GEMV).  Rank r then reads (b[r+1]^2 - b[r]^2)/2 ~ n^2/(2 world)
    elements per GEMV, whatever r."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank {rank} of {world}")
    if n % align or n // align < world:
        raise ValueError(f"n={n} must be a multiple of {align} with at least one strip per rank")
    nstrips = n // align
    bounds = [0]
    for r in range(1, world):
        b = int(round(nstrips * (r / world) ** 0.5))
        b = min(max(b, bounds[-1] + 1), nstrips - (world - r))  # strictly increasing, room for the others
        bounds.append(b)
    bounds.append(nstrips)
    return bounds[rank] * align, (bounds[rank + 1] - bounds[rank]) * align


def allreduce_in_place(gt, row0: int, nrows: int) -> None:
    """Symmetric schedule: every rank holds partial sums for all n entries; one ncclAllReduce(sum) adds them."""
    import torch.distributed as dist
    dist.all_reduce(gt)


def allgather_in_place(gt, row0: int, nrows: int) -> None:
    """Assemble the full gt on every rank: rank r contributes gt[row0:row0+nrows] (its own rows).
    With the nccl backend this is one ncclAllGather over xGMI, in place (send = recv + rank*count)."""
    import torch.distributed as dist
    dist.all_gather_into_tensor(gt, gt[row0:row0 + nrows])


class HipShardEngine:
    """One rank's row block in HBM, driven through the two-phase C ABI (include/ellhip.h)."""

    def __init__(self, n, row0, nrows, kappa, mq_rows, diag, xc, device=-1):
        import torch
        self._lib = capi.load()
        self.n, self.row0, self.nrows = n, row0, nrows
        h = C.c_void_p()
        capi.check(self._lib.ellhip_create_shard(C.byref(h), n, row0, nrows, float(kappa), _p(mq_rows), _p(diag),
                                                 _p(xc), device), "ellhip_create_shard")
        self.h = h
        self._torch = torch
        dev = torch.device("cuda", torch.cuda.current_device() if device < 0 else device)
        # gt lives in a tensor the collective library can address; all HIP work and the collective are
        # issued on one non-default torch stream, so they are ordered without host synchronisation
        self._gts = [torch.zeros(n, dtype=torch.float64, device=dev) for _ in range(2)]
        self._by_ptr = {t.data_ptr(): t for t in self._gts}
        self.stream = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize(dev)
        capi.check(self._lib.ellhip_set_gt_dev(h, C.c_void_p(self._gts[0].data_ptr()),
                                               C.c_void_p(self._gts[1].data_ptr())))
        capi.check(self._lib.ellhip_set_stream(h, C.c_void_p(self.stream.cuda_stream)))

    def issue(self):
        return self._torch.cuda.stream(self.stream)

    @property
    def gt(self):
        """The tensor holding Q*g of the most recently primed gradient (the one to all-gather)."""
        return self._by_ptr[self._lib.ellhip_gt_dev(self.h)]

    def begin(self, kind, g, b0, has1, b1):
        capi.check(self._lib.ellhip_update_begin(self.h, kind, _p(g), b0, has1, b1), "ellhip_update_begin")

    def end(self) -> int:
        return capi.check(self._lib.ellhip_update_end(self.h), "ellhip_update_end")

    def set_defer_depth(self, depth):
        capi.check(self._lib.ellhip_set_defer_depth(self.h, int(depth)), "ellhip_set_defer_depth")

    def flush(self):
        capi.check(self._lib.ellhip_flush(self.h), "ellhip_flush")

    def queue_primed(self) -> int:
        return int(self._lib.ellhip_queue_primed(self.h))

    def set_symmetric(self, flag=True):
        capi.check(self._lib.ellhip_set_shard_symmetric(self.h, int(flag)), "ellhip_set_shard_symmetric")

    def queue_upload(self, k, kinds, grads, b0, has1, b1):
        capi.check(self._lib.ellhip_queue_upload(self.h, k, _p(kinds), _p(grads), _p(b0), _p(has1), _p(b1)),
                   "ellhip_queue_upload")

    def queue_begin(self, i):
        capi.check(self._lib.ellhip_queue_begin(self.h, i), "ellhip_queue_begin")

    def queue_end(self, i):
        capi.check(self._lib.ellhip_queue_end(self.h, i), "ellhip_queue_end")

    def queue_prime(self, i):
        capi.check(self._lib.ellhip_queue_prime(self.h, i), "ellhip_queue_prime")

    def queue_cut(self, i):
        capi.check(self._lib.ellhip_queue_cut(self.h, i), "ellhip_queue_cut")

    def queue_commit(self, i, nxt):
        capi.check(self._lib.ellhip_queue_commit(self.h, i, nxt), "ellhip_queue_commit")

    def queue_results(self, k):
        st = np.empty(k, dtype=np.int32)
        ts = np.empty(k, dtype=np.float64)
        capi.check(self._lib.ellhip_queue_results(self.h, _p(st), _p(ts)), "ellhip_queue_results")
        return st, ts

    def xc(self):
        out = np.empty(self.n, dtype=np.float64)
        capi.check(self._lib.ellhip_get_xc(self.h, _p(out)))
        return out

    def set_xc(self, x):
        capi.check(self._lib.ellhip_set_xc(self.h, _p(x)))

    def mq_rows(self):
        out = np.empty((self.nrows, self.n), dtype=np.float64)
        capi.check(self._lib.ellhip_get_mq(self.h, _p(out)))
        return out

    def kappa(self):
        return self._lib.ellhip_kappa(self.h)

    def tsq(self):
        return self._lib.ellhip_tsq(self.h)

    def synchronize(self):
        capi.check(self._lib.ellhip_synchronize(self.h))

    def profile_enable(self, flag):
        capi.check(self._lib.ellhip_profile_enable(self.h, int(flag)))

    def profile_read(self):
        ms = np.zeros(capi.NKERNEL_CLASSES, dtype=np.float64)
        cnt = np.zeros(capi.NKERNEL_CLASSES, dtype=np.int64)
        capi.check(self._lib.ellhip_profile_read(self.h, _p(ms), _p(cnt)))
        return {name: (float(ms[i]), int(cnt[i])) for i, name in enumerate(capi.KERNEL_CLASS_NAMES)}

    def close(self):
        h, self.h = self.h, None
        if h:
            self._lib.ellhip_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedEll:
    """`Ell` (src/ell.rs) whose matrix is spread by row blocks over the ranks of a process group.
    Same SearchSpace surface as ellalgo_rs_amd.Ell; every rank must make the same calls."""

    def __init__(self, kappa, mq_rows, xc, *, diag=None, device=-1, rank=None, world=None,
                 engine_factory=None, exchange=None, symmetric=False, defer_depth=8):
        """symmetric=True: the deferred (depth 8) schedule with lower-triangle GEMVs and apply passes on row
        blocks of equal trapezoid area (partition_symmetric), partial sums added by ONE all-reduce per update:
        5 n^2 / P bytes per GPU and update instead of 9 n^2 / P.  Only that schedule is available then."""
        if rank is None or world is None:
            import torch.distributed as dist
            rank, world = dist.get_rank(), dist.get_world_size()
        self.rank, self.world = rank, world
        xc = _f64(xc)
        self.n = int(xc.size)
        self.symmetric = bool(symmetric)
        self.row0, self.nrows = (partition_symmetric if symmetric else partition)(self.n, world, rank)
        mq_rows = None if mq_rows is None else _f64(mq_rows, self.nrows * self.n)
        diag = None if diag is None else _f64(diag, self.n)
        factory = engine_factory or (lambda *a: HipShardEngine(*a, device=device))
        self.engine = factory(self.n, self.row0, self.nrows, float(kappa), mq_rows, diag, xc)
        self._exchange = exchange or (allreduce_in_place if symmetric else allgather_in_place)
        if symmetric:
            self.engine.set_symmetric(True)
            self.engine.set_defer_depth(defer_depth)
        self._qk = 0
        self._primed_index = -1

    @classmethod
    def new_with_scalar(cls, val, xc, **kw):
        return cls(float(val), None, xc, **kw)

    @classmethod
    def new(cls, val, xc, **kw):
        return cls(1.0, None, xc, diag=val, **kw)

    @classmethod
    def new_with_matrix(cls, kappa, mq_rows, xc, **kw):
        """mq_rows: THIS rank's row block (nrows x n) of a symmetric matrix."""
        return cls(kappa, mq_rows, xc, **kw)

    def _issue(self):
        return self.engine.issue() if hasattr(self.engine, "issue") else contextlib.nullcontext()

    # ---- SearchSpace
    def _update(self, kind, cut) -> CutStatus:
        grad, beta = cut
        g = _f64(grad, self.n)
        b0, has1, b1 = _split(beta)
        with self._issue():
            self.engine.begin(kind, g, b0, has1, b1)
            self._exchange(self.engine.gt, self.row0, self.nrows)
            return CutStatus(self.engine.end())

    def update_bias_cut(self, cut):
        return self._update(capi.CUT_BIAS, cut)

    def update_central_cut(self, cut):
        return self._update(capi.CUT_CENTRAL, cut)

    def update_q(self, cut):
        return self._update(capi.CUT_Q, cut)

    def xc(self):
        return self.engine.xc()

    def set_xc(self, x):
        self.engine.set_xc(_f64(x, self.n))

    def tsq(self):
        return self.engine.tsq()

    @property
    def kappa(self):
        return self.engine.kappa()

    def _sync_primed(self) -> None:
        """A shard whose recorded updates were applied by an observer (flush, mq_rows, a depth change) has dropped
        its primed GEMV -- it belonged to the old base: the next queue_run then primes, and exchanges, again."""
        if hasattr(self.engine, "queue_primed"):
            self._primed_index = self.engine.queue_primed()

    @property
    def mq_rows(self) -> np.ndarray:
        """This rank's row block of Q (symmetric mode: current up to each row's diagonal only)."""
        with self._issue():
            out = self.engine.mq_rows()
        self._sync_primed()
        return out

    def set_defer_depth(self, depth: int) -> None:
        """See ellhip_set_defer_depth: 1 = immediate shrink, 8 / 16 = recorded and applied in batches."""
        with self._issue():
            self.engine.set_defer_depth(depth)
        self._sync_primed()

    # ---- device-resident cut queue (every rank uploads the same cuts)
    def queue_upload(self, kinds, grads, beta0, beta1=None) -> int:
        grads = _f64(grads)
        k = grads.size // self.n
        kinds = np.ascontiguousarray(kinds, dtype=np.int32)
        beta0 = _f64(beta0, k)
        if beta1 is None:
            has1, b1 = np.zeros(k, dtype=np.int32), np.zeros(k, dtype=np.float64)
        else:
            b1 = _f64(beta1, k).copy()
            has1 = (~np.isnan(b1)).astype(np.int32)
            b1[np.isnan(b1)] = 0.0
        self.engine.queue_upload(k, kinds, grads, beta0, has1, b1)
        self._qk = k
        self._primed_index = -1
        return k

    def queue_run(self, first: int, count: int, fused: bool = False) -> None:
        eng, ex, row0, nrows = self.engine, self._exchange, self.row0, self.nrows
        with self._issue():
            if not fused:   # two passes over the local rows per cut
                for i in range(first, first + count):
                    eng.queue_begin(i)
                    ex(eng.gt, row0, nrows)
                    eng.queue_end(i)
                return
            # pipelined: one pass per cut; the collective follows whichever call ran a GEMV
            if self._primed_index != first:
                eng.queue_prime(first)
                ex(eng.gt, row0, nrows)
            for i in range(first, first + count):
                nxt = i + 1 if i + 1 < self._qk else -1
                eng.queue_cut(i)
                eng.queue_commit(i, nxt)
                if nxt >= 0:
                    ex(eng.gt, row0, nrows)
                self._primed_index = nxt

    def queue_results(self):
        st, ts = self.engine.queue_results(self._qk)
        if np.any(st[st >= 0] != 0):   # the queue halted: nothing stays primed
            self._primed_index = -1
        return st, ts

    def flush(self):
        """apply the recorded (deferred) updates now (local rows; no collective involved)"""
        with self._issue():
            self.engine.flush()
        # a shard whose recorded updates were applied has dropped its primed GEMV (it belonged to the old base):
        # the next queue_run primes -- and exchanges -- again
        self._sync_primed()

    def synchronize(self):
        self.engine.synchronize()

    def profile_enable(self, flag):
        self.engine.profile_enable(flag)

    def profile_read(self):
        return self.engine.profile_read()
