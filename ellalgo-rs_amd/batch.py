"""Python mirror of the batched small-n engine (include/ellhip_batch.h): B independent `Ell` search spaces
(src/ell.rs) of one dimension n <= 128 updated together, one workgroup per ellipsoid, bit-identical to the CPU
arithmetic.  Cuts are arrays over the batch: `grads[B][n]`, `beta0[B]`, `beta1[B]` (NaN = None)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .ell import _f64, _p


class EllBatch:
    def __init__(self, kappa, mq, xc, *, diag=None, device: int = -1, _handle=None):
        self._lib = capi.load()
        if _handle is None:
            xc = np.ascontiguousarray(xc, dtype=np.float64)
            if xc.ndim != 2:
                raise ValueError("xc must be [B][n]")
            B, n = xc.shape
            kappa = None if kappa is None else _f64(np.broadcast_to(np.asarray(kappa, dtype=np.float64), (B,)), B)
            mq = None if mq is None else _f64(mq, B * n * n)
            diag = None if diag is None else _f64(diag, B * n)
            h = C.c_void_p()
            capi.check(self._lib.ellhip_batch_create(C.byref(h), B, n, _p(kappa), _p(mq), _p(diag), _p(xc), device),
                       "ellhip_batch_create")
            _handle = h
        self._h = _handle
        self.B = int(self._lib.ellhip_batch_size(self._h))
        self.n = int(self._lib.ellhip_batch_ndim(self._h))

    # constructors, src/ell.rs:31-78 per ellipsoid
    @classmethod
    def new_with_matrix(cls, kappa, mq, xc, **kw):
        return cls(kappa, mq, xc, **kw)

    @classmethod
    def new(cls, val, xc, **kw):
        return cls(None, None, xc, diag=val, **kw)

    @classmethod
    def new_with_scalar(cls, val, xc, **kw):
        return cls(val, None, xc, **kw)

    @classmethod
    def from_space(cls, space, B: int):
        """B clones of one `Ell` (BSearchAdaptor's clone-per-probe, src/cutting_plane.rs:410)."""
        lib = capi.load()
        h = C.c_void_p()
        capi.check(lib.ellhip_batch_from_space(C.byref(h), space._h, int(B)), "ellhip_batch_from_space")
        return cls(None, None, None, _handle=h)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.ellhip_batch_destroy(h)

    def update(self, kinds, grads, beta0, beta1=None):
        """K cuts per ellipsoid.  kinds / beta0 / beta1: [K][B] (or [B] for K = 1), grads [K][B][n]; beta1 NaN or
        None = no second value.  Returns (status [K][B], tsq [K][B])."""
        grads = np.ascontiguousarray(grads, dtype=np.float64)
        if grads.ndim == 2:
            grads = grads[None]
        K = grads.shape[0]
        if grads.shape != (K, self.B, self.n):
            raise ValueError(f"grads must be [K][{self.B}][{self.n}]")
        kinds = np.ascontiguousarray(np.broadcast_to(np.asarray(kinds, dtype=np.int32), (K, self.B)))
        beta0 = np.ascontiguousarray(np.broadcast_to(np.asarray(beta0, dtype=np.float64), (K, self.B)))
        if beta1 is None:
            has1 = np.zeros((K, self.B), dtype=np.int32)
            b1 = np.zeros((K, self.B))
        else:
            b1 = np.ascontiguousarray(np.broadcast_to(np.asarray(beta1, dtype=np.float64), (K, self.B))).copy()
            has1 = np.ascontiguousarray((~np.isnan(b1)).astype(np.int32))
            b1[np.isnan(b1)] = 0.0
        status = np.empty((K, self.B), dtype=np.int32)
        tsq = np.empty((K, self.B), dtype=np.float64)
        capi.check(self._lib.ellhip_batch_update(self._h, K, _p(kinds), _p(grads), _p(beta0), _p(has1), _p(b1),
                                                 _p(status), _p(tsq)), "ellhip_batch_update")
        return status, tsq

    def update_dev(self, K, kinds_dev, grads_dev, beta0_dev, has1_dev, beta1_dev, status_dev, tsq_dev=None):
        """device pointers (ints), asynchronous on the handle's stream"""
        capi.check(self._lib.ellhip_batch_update_dev(self._h, int(K), kinds_dev, grads_dev, beta0_dev, has1_dev,
                                                     beta1_dev, status_dev, tsq_dev), "ellhip_batch_update_dev")

    def synchronize(self):
        capi.check(self._lib.ellhip_batch_synchronize(self._h))

    def _get(self, fn, shape):
        out = np.empty(shape, dtype=np.float64)
        capi.check(getattr(self._lib, fn)(self._h, _p(out)), fn)
        return out

    def xc(self):
        return self._get("ellhip_batch_get_xc", (self.B, self.n))

    def set_xc(self, x):
        x = _f64(x, self.B * self.n)
        capi.check(self._lib.ellhip_batch_set_xc(self._h, _p(x)), "ellhip_batch_set_xc")

    @property
    def mq(self):
        return self._get("ellhip_batch_get_mq", (self.B, self.n, self.n))

    @property
    def kappa(self):
        return self._get("ellhip_batch_get_kappa", (self.B,))

    def tsq(self):
        return self._get("ellhip_batch_get_tsq", (self.B,))

    def set_no_defer_trick(self, flag: bool):
        capi.check(self._lib.ellhip_batch_set_no_defer_trick(self._h, int(flag)))

    def set_use_parallel_cut(self, flag: bool):
        capi.check(self._lib.ellhip_batch_set_use_parallel_cut(self._h, int(flag)))
