"""ctypes mirror of include/ellhip_sharded.h: the row-partitioned Ell whose collective is issued by libellhip.so itself
(RCCL opened at run time), one process per GPU.  Test / bench plumbing -- a Rust or C++ host binds the same entry
points directly; `ellalgo_rs_amd.sharded.ShardedEll` is the older orchestration of the same
shard handle through torch.distributed."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .ell import CutStatus, _f64, _p, _split


def partition(n: int, nranks: int, rank: int, symmetric: bool = False):
    """(row0, nrows) of a rank, computed by the library (no device needed)."""
    r0, nr = C.c_int64(), C.c_int64()
    capi.check(capi.load().ellhip_sharded_partition(n, nranks, rank, int(symmetric), C.byref(r0), C.byref(nr)),
               "ellhip_sharded_partition")
    return int(r0.value), int(nr.value)


def unique_id() -> bytes:
    """ncclGetUniqueId through the library (rank 0 calls it and ships the bytes to the other ranks)."""
    buf = C.create_string_buffer(capi.NCCL_ID_BYTES)
    capi.check(capi.load().ellhip_sharded_unique_id(buf), "ellhip_sharded_unique_id")
    return buf.raw


class ShardedEllAbi:
    def __init__(self, kappa, mq_rows, xc, *, diag=None, device=-1, rank=0, nranks=1, nccl_id: bytes | None = None,
                 nccl_comm: int | None = None, symmetric=False, defer_depth=1):
        self._lib = capi.load()
        xc = _f64(xc)
        self.n = int(xc.size)
        self.rank, self.nranks, self.symmetric = rank, nranks, bool(symmetric)
        self.row0, self.nrows = partition(self.n, nranks, rank, symmetric)
        mq_rows = None if mq_rows is None else _f64(mq_rows, self.nrows * self.n)
        diag = None if diag is None else _f64(diag, self.n)
        idbuf = C.create_string_buffer(nccl_id, capi.NCCL_ID_BYTES) if nccl_id is not None else None
        h = C.c_void_p()
        capi.check(self._lib.ellhip_sharded_create(C.byref(h), self.n, float(kappa), _p(mq_rows), _p(diag), _p(xc), device,
                                                   rank, nranks, idbuf, C.c_void_p(nccl_comm) if nccl_comm else None,
                                                   int(symmetric), int(defer_depth)), "ellhip_sharded_create")
        self._h = h
        self._qk = 0

    @classmethod
    def new_with_scalar(cls, val, xc, **kw):
        return cls(float(val), None, xc, **kw)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.ellhip_sharded_destroy(h)
            except Exception:
                pass

    def _update(self, kind, cut) -> CutStatus:
        grad, beta = cut
        g = _f64(grad, self.n)
        b0, has1, b1 = _split(beta)
        return CutStatus(capi.check(self._lib.ellhip_sharded_update(self._h, kind, _p(g), b0, has1, b1),
                                    "ellhip_sharded_update"))

    def update_bias_cut(self, cut):
        return self._update(capi.CUT_BIAS, cut)

    def update_central_cut(self, cut):
        return self._update(capi.CUT_CENTRAL, cut)

    def update_q(self, cut):
        return self._update(capi.CUT_Q, cut)

    def xc(self):
        out = np.empty(self.n, dtype=np.float64)
        capi.check(self._lib.ellhip_sharded_get_xc(self._h, _p(out)), "ellhip_sharded_get_xc")
        return out

    def set_xc(self, x):
        capi.check(self._lib.ellhip_sharded_set_xc(self._h, _p(_f64(x, self.n))), "ellhip_sharded_set_xc")

    def tsq(self):
        return self._lib.ellhip_sharded_tsq(self._h)

    @property
    def kappa(self):
        return self._lib.ellhip_sharded_kappa(self._h)

    @property
    def mq_rows(self):
        out = np.empty((self.nrows, self.n), dtype=np.float64)
        capi.check(self._lib.ellhip_sharded_get_mq_rows(self._h, _p(out)), "ellhip_sharded_get_mq_rows")
        return out

    def set_defer_depth(self, depth):
        capi.check(self._lib.ellhip_sharded_set_defer_depth(self._h, int(depth)), "ellhip_sharded_set_defer_depth")

    def flush(self):
        capi.check(self._lib.ellhip_sharded_flush(self._h), "ellhip_sharded_flush")

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
        capi.check(self._lib.ellhip_sharded_queue_upload(self._h, k, _p(kinds), _p(grads), _p(beta0), _p(has1), _p(b1)),
                   "ellhip_sharded_queue_upload")
        self._qk = k
        return k

    def queue_run(self, first, count, fused=False):
        fn = self._lib.ellhip_sharded_queue_run_fused if fused else self._lib.ellhip_sharded_queue_run
        capi.check(fn(self._h, first, count), "ellhip_sharded_queue_run")

    def queue_results(self):
        st = np.empty(self._qk, dtype=np.int32)
        ts = np.empty(self._qk, dtype=np.float64)
        capi.check(self._lib.ellhip_sharded_queue_results(self._h, _p(st), _p(ts)), "ellhip_sharded_queue_results")
        return st, ts

    def synchronize(self):
        capi.check(self._lib.ellhip_sharded_synchronize(self._h), "ellhip_sharded_synchronize")

    def _local(self):
        return C.c_void_p(self._lib.ellhip_sharded_local(self._h))

    def set_local_option(self, key: int, value: int) -> None:
        """ellhip_set_option on this rank's shard handle (e.g. OPT_LOOKAHEAD = 1: the cut-by-cut schedule with one collective
        per update instead of one per group of queued cuts)"""
        capi.check(self._lib.ellhip_set_option(self._local(), int(key), int(value)), "ellhip_set_option")

    def get_local_option(self, key: int) -> int:
        v = C.c_int64()
        capi.check(self._lib.ellhip_get_option(self._local(), int(key), C.byref(v)), "ellhip_get_option")
        return int(v.value)

    def profile_enable(self, flag):
        capi.check(self._lib.ellhip_profile_enable(self._local(), int(flag)))

    def profile_read(self):
        ms = np.zeros(capi.NKERNEL_CLASSES, dtype=np.float64)
        cnt = np.zeros(capi.NKERNEL_CLASSES, dtype=np.int64)
        capi.check(self._lib.ellhip_profile_read(self._local(), _p(ms), _p(cnt)))
        return {name: (float(ms[i]), int(cnt[i])) for i, name in enumerate(capi.KERNEL_CLASS_NAMES)}
