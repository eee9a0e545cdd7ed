"""Row-block partitioned Ell over the GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).  Rank r owns rows
[r*n/P, (r+1)*n/P) of Q; every rank keeps full copies of g, gt = Q*g, xc and the scalars.  One
update is

    phase 1  local GEMV                gt[R_r] = Q[R_r, :] * g            (ellhip_*_begin)
    exchange in-place all-gather of gt (n/P doubles per rank)             (the ONLY collective)
    phase 2  redundant scalar stage (omega, tsq, EllCalc, xc, kappa: identical bits on every rank,
             fixed reduction shapes) + rank-1 update of the local rows    (ellhip_*_end)

The exchange primitive is injected (`exchange(gt_tensor, row0, nrows)`), so the orchestration is the
same code over RCCL on GPUs and in the world_size-2 gloo tests.  All HIP work and the collective
are issued on one non-default torch stream, so they are ordered without host synchronisation.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .ell import CutStatus, _SpaceBase, _f64, _p, _split


def partition(n: int, world: int, rank: int):
    """Contiguous equal row blocks; n must divide evenly so the all-gather is uniform."""
    if n % world:
        raise ValueError(f"n={n} is not divisible by world size {world}")
    nrows = n // world
    return rank * nrows, nrows


def _rccl_exchange(gt, row0, nrows):
    import torch.distributed as dist
    dist.all_gather_into_tensor(gt, gt[row0:row0 + nrows])


class ShardedEll(_SpaceBase):
    """`Ell` whose matrix is spread over the ranks of the default process group."""

    def __init__(self, kappa, mq_rows, xc, *, diag=None, device=-1, exchange=None, rank=None, world=None):
        import torch
        import torch.distributed as dist
        self._lib = capi.load()
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        xc = _f64(xc)
        self.n = int(xc.size)
        self.row0, self.nrows = partition(self.n, self.world, self.rank)
        mq_rows = None if mq_rows is None else _f64(mq_rows, self.nrows * self.n)
        diag = None if diag is None else _f64(diag, self.n)
        h = C.c_void_p()
        capi.check(self._lib.ellhip_create_shard(C.byref(h), self.n, self.row0, self.nrows, float(kappa),
                                                 _p(mq_rows), _p(diag), _p(xc), device), "ellhip_create_shard")
        self._h = h
        self._torch = torch
        dev = torch.device("cuda", torch.cuda.current_device() if device < 0 else device)
        # the gt buffer lives in a tensor the collective library can address
        self._gt = torch.zeros(self.n, dtype=torch.float64, device=dev)
        self._stream = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize(dev)
        capi.check(self._lib.ellhip_set_gt_dev(self._h, C.c_void_p(self._gt.data_ptr())))
        capi.check(self._lib.ellhip_set_stream(self._h, C.c_void_p(self._stream.cuda_stream)))
        self._exchange = exchange or _rccl_exchange

    @classmethod
    def new_with_scalar(cls, val, xc, **kw):
        return cls(float(val), None, xc, **kw)

    @classmethod
    def new(cls, val, xc, **kw):
        return cls(1.0, None, xc, diag=val, **kw)

    def clone(self):
        raise NotImplementedError("clone a sharded space rank by rank with ellhip_clone")

    def _update(self, kind, cut) -> CutStatus:
        grad, beta = cut
        g = _f64(grad, self.n)
        b0, has1, b1 = _split(beta)
        torch = self._torch
        with torch.cuda.stream(self._stream):
            capi.check(self._lib.ellhip_update_begin(self._h, kind, _p(g), b0, has1, b1), "ellhip_update_begin")
            self._exchange(self._gt, self.row0, self.nrows)
            return CutStatus(capi.check(self._lib.ellhip_update_end(self._h), "ellhip_update_end"))

    def queue_run(self, first: int, count: int) -> None:
        torch = self._torch
        lib, h = self._lib, self._h
        with torch.cuda.stream(self._stream):
            for i in range(first, first + count):
                capi.check(lib.ellhip_queue_begin(h, i), "ellhip_queue_begin")
                self._exchange(self._gt, self.row0, self.nrows)
                capi.check(lib.ellhip_queue_end(h, i), "ellhip_queue_end")

    @property
    def mq_rows(self) -> np.ndarray:
        """This rank's row block of Q."""
        out = np.empty((self.nrows, self.n), dtype=np.float64)
        capi.check(self._lib.ellhip_get_mq(self._h, _p(out)), "ellhip_get_mq")
        return out

    @property
    def mq(self):
        raise AttributeError("a sharded space holds only mq_rows; gather them on the host if needed")
