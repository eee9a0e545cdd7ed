"""Python mirror of the reference's search-space interface over the C ABI.

Same names, argument meaning and return values as `impl SearchSpace for Ell`
(src/ell.rs:140-180) and `for EllStable` (src/ell_stable.rs:128-166); all arithmetic happens in the
HIP engine behind include/ellhip.h.  A cut is `(grad, beta)` with `beta` a `SingleCut`, a
`ParallelCut`, a float (= SingleCut) or a 2-tuple (= ParallelCut).
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass
from typing import Optional, Tuple, Union

import numpy as np

from . import capi


class CutStatus(enum.IntEnum):
    """src/cutting_plane.rs:31-37 (declaration order)."""
    Success = 0
    NoSoln = 1
    NoEffect = 2
    Unknown = 3


@dataclass(frozen=True)
class SingleCut:
    """src/cutting_plane.rs:9"""
    beta: float


@dataclass(frozen=True)
class ParallelCut:
    """src/cutting_plane.rs:18"""
    beta0: float
    beta1: Optional[float] = None


CutChoice = Union[SingleCut, ParallelCut, float, Tuple[float, Optional[float]]]


def _split(beta: CutChoice):
    if isinstance(beta, SingleCut):
        return float(beta.beta), 0, 0.0
    if isinstance(beta, ParallelCut):
        return float(beta.beta0), int(beta.beta1 is not None), 0.0 if beta.beta1 is None else float(beta.beta1)
    if isinstance(beta, (tuple, list)):
        b0, b1 = beta
        return float(b0), int(b1 is not None), 0.0 if b1 is None else float(b1)
    return float(beta), 0, 0.0


def _f64(a, size=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if size is not None and a.size != size:
        raise ValueError(f"expected {size} elements, got {a.size}")
    return a


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class _SpaceBase:
    _variant = capi.SPACE_ELL

    def __init__(self, kappa: float, mq, xc, *, diag=None, device: int = -1, _handle=None, _n=None):
        self._lib = capi.load()
        if _handle is not None:
            self._h, self.n = _handle, _n
            return
        xc = _f64(xc)
        self.n = int(xc.size)
        mq = None if mq is None else _f64(mq, self.n * self.n)
        diag = None if diag is None else _f64(diag, self.n)
        h = C.c_void_p()
        capi.check(self._lib.ellhip_create(C.byref(h), self._variant, self.n, float(kappa), _p(mq), _p(diag),
                                           _p(xc), device), "ellhip_create")
        self._h = h

    # ---- constructors, src/ell.rs:31-78 / src/ell_stable.rs:18-35
    @classmethod
    def new_with_matrix(cls, kappa, mq, xc, **kw):
        return cls(kappa, mq, xc, **kw)

    @classmethod
    def new(cls, val, xc, **kw):
        return cls(1.0, None, xc, diag=val, **kw)

    @classmethod
    def new_with_scalar(cls, val, xc, **kw):
        return cls(float(val), None, xc, **kw)

    def clone(self):
        h = C.c_void_p()
        capi.check(self._lib.ellhip_clone(self._h, C.byref(h)), "ellhip_clone")
        return type(self)(0.0, None, None, _handle=h, _n=self.n)

    __copy__ = clone

    def __deepcopy__(self, memo):
        return self.clone()

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.ellhip_destroy(h)
            except Exception:
                pass

    # ---- SearchSpace
    def xc(self) -> np.ndarray:
        out = np.empty(self.n, dtype=np.float64)
        capi.check(self._lib.ellhip_get_xc(self._h, _p(out)), "ellhip_get_xc")
        return out

    def tsq(self) -> float:
        return self._lib.ellhip_tsq(self._h)

    def set_xc(self, x) -> None:
        x = _f64(x, self.n)
        capi.check(self._lib.ellhip_set_xc(self._h, _p(x)), "ellhip_set_xc")

    def _update(self, kind, cut) -> CutStatus:
        grad, beta = cut
        g = _f64(grad, self.n)
        b0, has1, b1 = _split(beta)
        return CutStatus(capi.check(self._lib.ellhip_update(self._h, kind, _p(g), b0, has1, b1), "ellhip_update"))

    def update_bias_cut(self, cut) -> CutStatus:
        return self._update(capi.CUT_BIAS, cut)

    def update_central_cut(self, cut) -> CutStatus:
        return self._update(capi.CUT_CENTRAL, cut)

    def update_q(self, cut) -> CutStatus:
        return self._update(capi.CUT_Q, cut)

    # ---- pipelined form (include/ellhip.h "pipelined update"): same results, one pass over Q per update
    def prime(self, grad) -> None:
        g = _f64(grad, self.n)
        capi.check(self._lib.ellhip_prime(self._h, _p(g)), "ellhip_prime")

    def cut(self, kind: int, beta) -> CutStatus:
        b0, has1, b1 = _split(beta)
        return CutStatus(capi.check(self._lib.ellhip_cut(self._h, kind, b0, has1, b1), "ellhip_cut"))

    def commit(self, next_grad=None) -> None:
        g = None if next_grad is None else _f64(next_grad, self.n)
        capi.check(self._lib.ellhip_commit(self._h, _p(g)), "ellhip_commit")

    # ---- fields
    @property
    def kappa(self) -> float:
        return self._lib.ellhip_kappa(self._h)

    @property
    def mq(self) -> np.ndarray:
        out = np.empty((self.n, self.n), dtype=np.float64)
        capi.check(self._lib.ellhip_get_mq(self._h, _p(out)), "ellhip_get_mq")
        return out

    def set_use_parallel_cut(self, flag: bool) -> None:
        capi.check(self._lib.ellhip_set_use_parallel_cut(self._h, int(flag)))

    def set_option(self, key: int, value: int) -> None:
        """ellhip_set_option (keys: capi.OPT_*)"""
        capi.check(self._lib.ellhip_set_option(self._h, int(key), int(value)), "ellhip_set_option")

    def get_option(self, key: int) -> int:
        v = C.c_int64()
        capi.check(self._lib.ellhip_get_option(self._h, int(key), C.byref(v)), "ellhip_get_option")
        return int(v.value)

    # ---- device-resident cut queue
    def queue_upload(self, kinds, grads, beta0, beta1=None) -> int:
        """beta1: array with NaN where the cut has no second value, or None."""
        grads = _f64(grads)
        k = grads.size // self.n
        kinds = np.ascontiguousarray(kinds, dtype=np.int32)
        beta0 = _f64(beta0, k)
        if beta1 is None:
            has1 = np.zeros(k, dtype=np.int32)
            b1 = np.zeros(k, dtype=np.float64)
        else:
            b1 = _f64(beta1, k).copy()
            has1 = (~np.isnan(b1)).astype(np.int32)
            b1[np.isnan(b1)] = 0.0
        capi.check(self._lib.ellhip_queue_upload(self._h, k, _p(kinds), _p(grads), _p(beta0), _p(has1), _p(b1)),
                   "ellhip_queue_upload")
        self._qk = k
        return k

    def queue_run(self, first: int, count: int, fused: bool = False) -> None:
        fn = self._lib.ellhip_queue_run_fused if fused else self._lib.ellhip_queue_run
        capi.check(fn(self._h, first, count), "ellhip_queue_run")

    def queue_results(self):
        st = np.empty(self._qk, dtype=np.int32)
        ts = np.empty(self._qk, dtype=np.float64)
        capi.check(self._lib.ellhip_queue_results(self._h, _p(st), _p(ts)), "ellhip_queue_results")
        return st, ts

    def synchronize(self) -> None:
        capi.check(self._lib.ellhip_synchronize(self._h))

    def set_stream(self, stream_ptr: int) -> None:
        capi.check(self._lib.ellhip_set_stream(self._h, C.c_void_p(stream_ptr)))

    def flush(self) -> None:
        """ellhip_flush: apply the recorded (deferred) updates now"""
        capi.check(self._lib.ellhip_flush(self._h), "ellhip_flush")

    def profile_enable(self, flag: bool) -> None:
        capi.check(self._lib.ellhip_profile_enable(self._h, int(flag)))

    def profile_read(self):
        ms = np.zeros(capi.NKERNEL_CLASSES, dtype=np.float64)
        cnt = np.zeros(capi.NKERNEL_CLASSES, dtype=np.int64)
        capi.check(self._lib.ellhip_profile_read(self._h, _p(ms), _p(cnt)))
        return {name: (float(ms[i]), int(cnt[i])) for i, name in enumerate(capi.KERNEL_CLASS_NAMES)}


class Ell(_SpaceBase):
    """`Ell` (src/ell.rs:9-16) on the GPU."""
    _variant = capi.SPACE_ELL

    @classmethod
    def from_covariance(cls, cov, xc, **kw):
        return cls(1.0, cov, xc, **kw)

    @property
    def no_defer_trick(self) -> bool:
        return getattr(self, "_ndt", False)

    @no_defer_trick.setter
    def no_defer_trick(self, flag: bool) -> None:
        capi.check(self._lib.ellhip_set_no_defer_trick(self._h, int(flag)))
        self._ndt = bool(flag)

    @property
    def defer_depth(self) -> int:
        """1 = shrink Q at every cut (reference data flow); 8 / 16 = record cuts and apply them in batches.
        A new handle with even n >= 5120 starts at 24, another one with n >= 3072 at 8 (include/ellhip.h, "deferred
        shrink"), anything smaller at 1."""
        return self._lib.ellhip_defer_depth(self._h)

    @defer_depth.setter
    def defer_depth(self, depth: int) -> None:
        capi.check(self._lib.ellhip_set_defer_depth(self._h, int(depth)), "ellhip_set_defer_depth")

    def clone(self):
        c = super().clone()
        c._ndt = self.no_defer_trick
        return c


class EllStable(_SpaceBase):
    """`EllStable` (src/ell_stable.rs:9-15) on the GPU."""
    _variant = capi.SPACE_ELL_STABLE


def calc(n: int, kind: int, beta: CutChoice, tsq: float, use_parallel_cut: bool = True, device: int = -1):
    """EllCalc on the device: returns (CutStatus, (rho, sigma, delta))."""
    b0, has1, b1 = _split(beta)
    out = (C.c_double * 3)()
    st = capi.check(capi.load().ellhip_calc(n, int(use_parallel_cut), kind, b0, has1, b1, float(tsq), out, device),
                    "ellhip_calc")
    return CutStatus(st), (out[0], out[1], out[2])
