"""Python mirror of `LDLTMgr` (src/oracles/ldlt_mgr.rs), `LMIOracle` (src/oracles/lmi_oracle.rs) and `LMI0Oracle`
(src/oracles/lmi0_oracle.rs) over the C ABI of include/ellhip_lmi.h: the matrices and the factorisation live on
the GPU.  Same method names and return shapes as the reference."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import capi
from .ell import SingleCut, _f64, _p


class _LmiHandle:
    def __init__(self, mat_f, mat_b, device):
        self._lib = capi.load()
        if mat_f is None:
            self.n = 0
            mb = np.ascontiguousarray(mat_b, dtype=np.float64)
            self.m = int(mb.shape[0])
            mf = None
        else:
            mf = np.ascontiguousarray(mat_f, dtype=np.float64)
            if mf.ndim != 3 or mf.shape[1] != mf.shape[2]:
                raise ValueError("mat_f must be [n][m][m]")
            self.n, self.m = int(mf.shape[0]), int(mf.shape[1])
            mb = None if mat_b is None else _f64(mat_b, self.m * self.m)
        h = C.c_void_p()
        capi.check(self._lib.ellhip_lmi_create(C.byref(h), self.n, self.m, _p(mf), _p(mb), device), "ellhip_lmi_create")
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.ellhip_lmi_destroy(h)

    def _assess(self, x):
        g = np.empty(max(self.n, 1), dtype=np.float64)
        ep = C.c_double()
        xx = None if self.n == 0 else _f64(x, self.n)
        rc = capi.check(self._lib.ellhip_lmi_assess_feas(self._h, _p(xx), _p(g), C.byref(ep)), "ellhip_lmi_assess_feas")
        return rc, g[:self.n], ep.value

    @property
    def pos(self) -> Tuple[int, int]:
        p = np.zeros(2, dtype=np.int64)
        capi.check(self._lib.ellhip_lmi_pos(self._h, _p(p)), "ellhip_lmi_pos")
        return int(p[0]), int(p[1])

    @property
    def wit(self) -> np.ndarray:
        out = np.empty(self.m, dtype=np.float64)
        capi.check(self._lib.ellhip_lmi_get_witness(self._h, _p(out)), "ellhip_lmi_get_witness")
        return out

    @property
    def storage(self) -> np.ndarray:
        out = np.empty((self.m, self.m), dtype=np.float64)
        capi.check(self._lib.ellhip_lmi_get_storage(self._h, _p(out)), "ellhip_lmi_get_storage")
        return out


class LDLTMgr:
    """src/oracles/ldlt_mgr.rs: factorize / is_spd / witness / sqrt (the lazy `factor` closure form is what
    LMIOracle uses internally)."""

    def __init__(self, ndim: int, *, device: int = -1):
        self.ndim = int(ndim)
        self._device = device
        self._o: Optional[_LmiHandle] = None
        self._ep = 0.0

    def factorize(self, mat) -> bool:
        mat = np.ascontiguousarray(mat, dtype=np.float64)
        if mat.shape != (self.ndim, self.ndim):
            raise ValueError("matrix must be ndim x ndim")
        self._o = _LmiHandle(None, mat, self._device)
        rc, _, self._ep = self._o._assess(None)
        return rc == 0

    def is_spd(self) -> bool:
        return self.pos[1] == 0

    @property
    def pos(self):
        return (0, 0) if self._o is None else self._o.pos

    @property
    def wit(self):
        return np.zeros(self.ndim) if self._o is None else self._o.wit

    @property
    def storage(self):
        return np.zeros((self.ndim, self.ndim)) if self._o is None else self._o.storage

    def witness(self) -> float:
        assert not self.is_spd(), "witness called on SPD matrix"
        return self._ep

    def sqrt(self) -> np.ndarray:
        assert self._o is not None and self.is_spd(), "sqrt called on non-SPD matrix"
        out = np.empty((self.ndim, self.ndim), dtype=np.float64)
        capi.check(self._o._lib.ellhip_lmi_sqrt(self._o._h, _p(out)), "ellhip_lmi_sqrt")
        return out


class LMIOracle(_LmiHandle):
    """src/oracles/lmi_oracle.rs: F(x) = B - sum x_k F_k > 0"""

    def __init__(self, mat_f, mat_b, *, device: int = -1):
        super().__init__(mat_f, mat_b, device)

    def assess_feas(self, xc) -> Optional[Tuple[np.ndarray, SingleCut]]:
        rc, g, ep = self._assess(xc)
        return None if rc == 0 else (g.copy(), SingleCut(ep))


class LMI0Oracle(_LmiHandle):
    """src/oracles/lmi0_oracle.rs: F(x) = sum x_k F_k > 0; returns (g, ep) with a plain float like the reference"""

    def __init__(self, mat_f, *, device: int = -1):
        super().__init__(mat_f, None, device)

    def assess_feas(self, x) -> Optional[Tuple[np.ndarray, float]]:
        rc, g, ep = self._assess(x)
        return None if rc == 0 else (g.copy(), ep)
