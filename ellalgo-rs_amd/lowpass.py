"""Python mirror of `LowpassOracle` (src/oracles/lowpass_oracle.rs:7-167) over the C ABI of
include/ellhip_lowpass.h: the table and the walk live on the GPU.  Same method names and return
shapes as the reference's `OracleFeas` / `OracleOptim` impls; `cutting_plane_optim` /
`cutting_plane_feas` run the reference's driver loops (src/cutting_plane.rs:205-227,286-313) entirely on
the device with an `Ell` / `EllStable` from ell.py as the search space.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Tuple

import numpy as np

from . import capi
from .ell import ParallelCut, _f64, _p


def lowpass_case_constants(corrected: bool = False):
    """(wpass, wstop, lp_sq, up_sq, sp_sq) of create_lowpass_case (:153-167).  As written they give
    lp_sq > up_sq (every run ends NoSoln at iteration 0); corrected=True uses the ripple the constants'
    names describe: 20 log10(1 + 0.025) for the passband, 20 log10(0.125) for the stopband."""
    d0p, d0s = 0.025, 0.125
    if corrected:
        delta1 = 20.0 * math.log10(1.0 + d0p)
        delta2 = 20.0 * math.log10(d0s)
    else:
        delta1 = 20.0 * math.log10(d0p * math.pi)
        delta2 = 20.0 * math.log10(d0s * math.pi)
    low_pass = math.pow(10.0, -delta1 / 20.0)
    up_pass = math.pow(10.0, delta1 / 20.0)
    stop_pass = math.pow(10.0, delta2 / 20.0)
    return 0.12, 0.20, low_pass * low_pass, up_pass * up_pass, stop_pass * stop_pass


class LowpassOracle:
    def __init__(self, ndim: int, wpass: float, wstop: float, lp_sq: float, up_sq: float, sp_sq: float, *,
                 spectrum=None, device: int = -1):
        self._lib = capi.load()
        self.n = int(ndim)
        spec = None if spectrum is None else _f64(spectrum, 15 * self.n * self.n)
        h = C.c_void_p()
        capi.check(self._lib.ellhip_lowpass_create(C.byref(h), self.n, float(wpass), float(wstop), float(lp_sq),
                                                   float(up_sq), float(sp_sq), _p(spec), device),
                   "ellhip_lowpass_create")
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.ellhip_lowpass_destroy(h)

    # ---- public fields of the reference struct
    def state(self) -> dict:
        ints = np.zeros(7, dtype=np.int32)
        dbl = np.zeros(2, dtype=np.float64)
        capi.check(self._lib.ellhip_lowpass_state(self._h, _p(ints), _p(dbl)), "ellhip_lowpass_state")
        return dict(more_alt=int(ints[0]), idx1=int(ints[1]), idx2=int(ints[2]), idx3=int(ints[3]), kmax=int(ints[4]),
                    nwpass=int(ints[5]), nwstop=int(ints[6]), fmax=float(dbl[0]), sp_sq=float(dbl[1]))

    def rows_visited(self, reset: bool = False) -> int:
        """rows the reference's walk visits, summed over calls (work measure)"""
        return int(capi.check(self._lib.ellhip_lowpass_rows_visited(self._h, int(reset)), "ellhip_lowpass_rows_visited"))

    @property
    def spectrum(self) -> np.ndarray:
        out = np.empty((15 * self.n, self.n), dtype=np.float64)
        capi.check(self._lib.ellhip_lowpass_get_spectrum(self._h, _p(out)), "ellhip_lowpass_get_spectrum")
        return out

    # ---- OracleFeas / OracleOptim
    def assess_feas(self, x) -> Optional[Tuple[np.ndarray, ParallelCut]]:
        x = _f64(x, self.n)
        g = np.empty(self.n, dtype=np.float64)
        b0, b1, h = C.c_double(), C.c_double(), C.c_int()
        rc = capi.check(self._lib.ellhip_lowpass_assess_feas(self._h, _p(x), _p(g), C.byref(b0), C.byref(h),
                                                             C.byref(b1)), "ellhip_lowpass_assess_feas")
        if rc == 0:
            return None
        return g, ParallelCut(b0.value, b1.value if h.value else None)

    def assess_optim(self, x, gamma: float):
        """((grad, ParallelCut), shrunk, gamma) -- gamma is the reference's `&mut sp_sq`"""
        x = _f64(x, self.n)
        g = np.empty(self.n, dtype=np.float64)
        b0, b1, h, sh, gm = C.c_double(), C.c_double(), C.c_int(), C.c_int(), C.c_double(gamma)
        capi.check(self._lib.ellhip_lowpass_assess_optim(self._h, _p(x), C.byref(gm), _p(g), C.byref(b0), C.byref(h),
                                                         C.byref(b1), C.byref(sh)), "ellhip_lowpass_assess_optim")
        return (g, ParallelCut(b0.value, b1.value if h.value else None)), bool(sh.value), gm.value

    # ---- device-resident driver loops
    def cutting_plane_optim(self, space, gamma: float, max_iters: int, tol: float):
        """(x_best or None, niter, gamma)"""
        xb = np.empty(self.n, dtype=np.float64)
        hb, ni, gm = C.c_int(), C.c_int64(), C.c_double(gamma)
        capi.check(self._lib.ellhip_lowpass_optim(space._h, self._h, C.byref(gm), int(max_iters), float(tol), _p(xb),
                                                  C.byref(hb), C.byref(ni)), "ellhip_lowpass_optim")
        return (xb if hb.value else None), int(ni.value), gm.value

    def cutting_plane_feas(self, space, max_iters: int, tol: float):
        """(x or None, niter)"""
        xb = np.empty(self.n, dtype=np.float64)
        hb, ni = C.c_int(), C.c_int64()
        capi.check(self._lib.ellhip_lowpass_feas(space._h, self._h, int(max_iters), float(tol), _p(xb), C.byref(hb),
                                                 C.byref(ni)), "ellhip_lowpass_feas")
        return (xb if hb.value else None), int(ni.value)


def create_lowpass_case(ndim: int, *, corrected: bool = False, device: int = -1) -> LowpassOracle:
    """src/oracles/lowpass_oracle.rs:153-167"""
    return LowpassOracle(ndim, *lowpass_case_constants(corrected), device=device)
