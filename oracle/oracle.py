"""ctypes view of the CPU oracle (oracle/ell_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; nothing under ``ellalgo-rs_amd/`` does.  It never touches /root/reference.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libell_oracle.so")

SUCCESS, NOSOLN, NOEFFECT, UNKNOWN = 0, 1, 2, 3
CUT_BIAS, CUT_CENTRAL, CUT_Q = 0, 1, 2


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("ell_oracle.c", "ell_oracle.h", "lowpass_oracle.c", "lowpass_oracle.h", "lmi_oracle.c",
                                              "lmi_oracle.h")]
    stale = (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libell_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Calc(C.Structure):
    _fields_ = [("n_f", C.c_double), ("n_plus_1", C.c_double), ("half_n", C.c_double),
                ("inv_n", C.c_double), ("cst1", C.c_double), ("cst2", C.c_double),
                ("use_parallel_cut", C.c_int)]


class _Lowpass(C.Structure):
    _fields_ = [("more_alt", C.c_int), ("idx1", C.c_int), ("spectrum", C.POINTER(C.c_double)),
                ("ndim", C.c_int64), ("mdim", C.c_int64), ("nwpass", C.c_int), ("nwstop", C.c_int),
                ("lp_sq", C.c_double), ("up_sq", C.c_double), ("sp_sq", C.c_double),
                ("idx2", C.c_int), ("idx3", C.c_int), ("fmax", C.c_double), ("kmax", C.c_int),
                ("rows_visited", C.c_int64)]


class _Ldlt(C.Structure):
    _fields_ = [("pos0", C.c_int64), ("pos1", C.c_int64), ("wit", C.POINTER(C.c_double)), ("ndim", C.c_int64),
                ("storage", C.POINTER(C.c_double))]


class _Lmi(C.Structure):
    _fields_ = [("mode", C.c_int), ("n", C.c_int64), ("m", C.c_int64), ("mat_f", C.POINTER(C.c_double)),
                ("mat_b", C.POINTER(C.c_double)), ("ldlt", C.POINTER(_Ldlt))]


_ELEM_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_int64, C.c_int64)

_lib = None
_dp = C.POINTER(C.c_double)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        out3 = C.c_double * 3
        L.orc_calc_init.argtypes = [C.POINTER(_Calc), C.c_int64]
        for name, nargs in [("orc_calc_bias_cut", 2), ("orc_calc_bias_cut_q", 2), ("orc_calc_central_cut", 1),
                            ("orc_calc_parallel_bias_cut", 3), ("orc_calc_parallel_q", 3),
                            ("orc_calc_parallel_central_cut", 2)]:
            f = getattr(L, name)
            f.argtypes = [C.POINTER(_Calc)] + [C.c_double] * nargs + [out3]
            f.restype = C.c_int
        for name, nargs in [("orc_core_parallel_bias_cut_fast", 5), ("orc_core_parallel_bias_cut", 3),
                            ("orc_core_parallel_central_cut", 2), ("orc_core_bias_cut_fast", 3),
                            ("orc_core_bias_cut", 2), ("orc_core_central_cut", 1)]:
            f = getattr(L, name)
            f.argtypes = [C.POINTER(_Calc)] + [C.c_double] * nargs + [out3]
            f.restype = None
        L.orc_calc_dispatch.argtypes = [C.POINTER(_Calc), C.c_int, C.c_double, C.c_int, C.c_double,
                                        C.c_double, out3]
        L.orc_calc_dispatch.restype = C.c_int
        for pre in ("orc_ell", "orc_ellstable"):
            getattr(L, pre + "_new").argtypes = [C.c_int64, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
            getattr(L, pre + "_new").restype = C.c_void_p
            getattr(L, pre + "_clone").argtypes = [C.c_void_p]
            getattr(L, pre + "_clone").restype = C.c_void_p
            getattr(L, pre + "_free").argtypes = [C.c_void_p]
            getattr(L, pre + "_free").restype = None
            getattr(L, pre + "_update").argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_int, C.c_double]
            getattr(L, pre + "_update").restype = C.c_int
            getattr(L, pre + "_kappa").argtypes = [C.c_void_p]
            getattr(L, pre + "_kappa").restype = C.c_double
            getattr(L, pre + "_tsq").argtypes = [C.c_void_p]
            getattr(L, pre + "_tsq").restype = C.c_double
            getattr(L, pre + "_mq").argtypes = [C.c_void_p]
            getattr(L, pre + "_mq").restype = _dp
            getattr(L, pre + "_xc").argtypes = [C.c_void_p]
            getattr(L, pre + "_xc").restype = _dp
        L.orc_ell_update_rowwise.argtypes = L.orc_ell_update.argtypes
        L.orc_ell_update_rowwise.restype = C.c_int
        L.orc_ell_update_rowwise_mt.argtypes = L.orc_ell_update.argtypes
        L.orc_ell_update_rowwise_mt.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_set_num_threads.restype = C.c_int
        L.orc_ell_set_no_defer_trick.argtypes = [C.c_void_p, C.c_int]
        L.orc_ell_set_use_parallel_cut.argtypes = [C.c_void_p, C.c_int]
        L.orc_ellstable_set_corrected.argtypes = [C.c_void_p, C.c_int]
        L.orc_rows_gemv.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_rows_gemv.restype = None
        L.orc_ell_batch_run.argtypes = [C.c_int64, C.c_int64, C.c_int64] + [C.c_void_p] * 5 + [C.c_double] + [C.c_void_p] * 4
        L.orc_ell_batch_run.restype = C.c_int64
        L.orc_ldlt_new.argtypes = [C.c_int64]
        L.orc_ldlt_new.restype = C.POINTER(_Ldlt)
        L.orc_ldlt_free.argtypes = [C.POINTER(_Ldlt)]
        L.orc_ldlt_free.restype = None
        for nm in ("orc_ldlt_factor", "orc_ldlt_factor_semidefinite"):
            getattr(L, nm).argtypes = [C.POINTER(_Ldlt), _ELEM_FN, C.c_void_p]
            getattr(L, nm).restype = C.c_int
        L.orc_ldlt_factorize.argtypes = [C.POINTER(_Ldlt), C.c_void_p]
        L.orc_ldlt_factorize.restype = C.c_int
        L.orc_ldlt_witness.argtypes = [C.POINTER(_Ldlt)]
        L.orc_ldlt_witness.restype = C.c_double
        L.orc_ldlt_sym_quad.argtypes = [C.POINTER(_Ldlt), C.c_void_p]
        L.orc_ldlt_sym_quad.restype = C.c_double
        L.orc_ldlt_sqrt.argtypes = [C.POINTER(_Ldlt), C.c_void_p]
        L.orc_ldlt_sqrt.restype = None
        L.orc_lmi_new.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
        L.orc_lmi_new.restype = C.POINTER(_Lmi)
        L.orc_lmi_free.argtypes = [C.POINTER(_Lmi)]
        L.orc_lmi_free.restype = None
        L.orc_lmi_assess_feas.argtypes = [C.POINTER(_Lmi), C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.orc_lmi_assess_feas.restype = C.c_int
        L.orc_lowpass_new.argtypes = [C.c_int64] + [C.c_double] * 5
        L.orc_lowpass_new.restype = C.POINTER(_Lowpass)
        L.orc_lowpass_free.argtypes = [C.POINTER(_Lowpass)]
        L.orc_lowpass_free.restype = None
        L.orc_lowpass_case.argtypes = [C.c_int, C.c_double * 5]
        L.orc_lowpass_case.restype = None
        ip = C.POINTER(C.c_int)
        L.orc_lowpass_assess_feas.argtypes = [C.POINTER(_Lowpass), C.c_void_p, C.c_void_p, _dp, ip, _dp]
        L.orc_lowpass_assess_feas.restype = C.c_int
        L.orc_lowpass_assess_optim.argtypes = [C.POINTER(_Lowpass), C.c_void_p, _dp, C.c_void_p, _dp, ip, _dp, ip]
        L.orc_lowpass_assess_optim.restype = C.c_int
        L.orc_lowpass_cutting_plane_optim.argtypes = [C.POINTER(_Lowpass), C.c_int, C.c_void_p, _dp, C.c_int64,
                                                      C.c_double, C.c_void_p, ip, ip]
        L.orc_lowpass_cutting_plane_optim.restype = C.c_int64
        L.orc_lowpass_cutting_plane_feas.argtypes = [C.POINTER(_Lowpass), C.c_int, C.c_void_p, C.c_int64, C.c_double,
                                                     C.c_void_p, ip, ip]
        L.orc_lowpass_cutting_plane_feas.restype = C.c_int64
        _lib = L
    return _lib


def _arr(a, n=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float64)
    if n is not None:
        assert a.size == n, (a.size, n)
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Calc:
    """EllCalc (src/ell_calc.rs:627-931) as restated by the oracle."""

    def __init__(self, n: int):
        self.c = _Calc()
        lib().orc_calc_init(C.byref(self.c), n)

    def _gate(self, name, *args):
        out = (C.c_double * 3)()
        st = getattr(lib(), name)(C.byref(self.c), *args, out)
        return st, (out[0], out[1], out[2])

    def _core(self, name, *args):
        out = (C.c_double * 3)()
        getattr(lib(), name)(C.byref(self.c), *args, out)
        return (out[0], out[1], out[2])

    def calc_bias_cut(self, beta, tsq): return self._gate("orc_calc_bias_cut", beta, tsq)
    def calc_bias_cut_q(self, beta, tsq): return self._gate("orc_calc_bias_cut_q", beta, tsq)
    def calc_central_cut(self, tsq): return self._gate("orc_calc_central_cut", tsq)
    def calc_parallel_bias_cut(self, b0, b1, tsq): return self._gate("orc_calc_parallel_bias_cut", b0, b1, tsq)
    def calc_parallel_q(self, b0, b1, tsq): return self._gate("orc_calc_parallel_q", b0, b1, tsq)
    def calc_parallel_central_cut(self, b1, tsq): return self._gate("orc_calc_parallel_central_cut", b1, tsq)

    def core_parallel_bias_cut_fast(self, b0, b1, tsq, b0b1, eta):
        return self._core("orc_core_parallel_bias_cut_fast", b0, b1, tsq, b0b1, eta)

    def core_bias_cut_fast(self, beta, tau, eta): return self._core("orc_core_bias_cut_fast", beta, tau, eta)

    def dispatch(self, kind, b0, b1, tsq):
        """b1 is None for SingleCut / ParallelCut(b0, None)."""
        out = (C.c_double * 3)()
        st = lib().orc_calc_dispatch(C.byref(self.c), kind, b0, int(b1 is not None),
                                     0.0 if b1 is None else b1, tsq, out)
        return st, (out[0], out[1], out[2])


class _Space:
    _pre = ""

    def __init__(self, n, kappa=1.0, mq=None, diag=None, xc=None, _handle=None):
        self.n = int(n)
        if _handle is not None:
            self.h = _handle
            return
        mq = _arr(mq, self.n * self.n)
        diag = _arr(diag, self.n)
        xc = _arr(xc, self.n)
        self.h = getattr(lib(), self._pre + "_new")(self.n, float(kappa), _ptr(mq), _ptr(diag), _ptr(xc))

    # reference-style constructors (src/ell.rs:31-78, src/ell_stable.rs:18-35)
    @classmethod
    def new_with_scalar(cls, val, xc):
        xc = np.asarray(xc, dtype=np.float64)
        return cls(xc.size, kappa=val, xc=xc)

    @classmethod
    def new(cls, val, xc):
        xc = np.asarray(xc, dtype=np.float64)
        return cls(xc.size, kappa=1.0, diag=val, xc=xc)

    @classmethod
    def new_with_matrix(cls, kappa, mq, xc):
        xc = np.asarray(xc, dtype=np.float64)
        return cls(xc.size, kappa=kappa, mq=mq, xc=xc)

    def clone(self):
        return type(self)(self.n, _handle=getattr(lib(), self._pre + "_clone")(self.h))

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h and _lib is not None:
            getattr(_lib, self._pre + "_free")(h)

    def update(self, kind, grad, b0, b1=None):
        g = _arr(grad, self.n)
        return getattr(lib(), self._pre + "_update")(self.h, kind, _ptr(g), float(b0), int(b1 is not None),
                                                      0.0 if b1 is None else float(b1))

    def update_bias_cut(self, grad, b0, b1=None): return self.update(CUT_BIAS, grad, b0, b1)
    def update_central_cut(self, grad, b0=0.0, b1=None): return self.update(CUT_CENTRAL, grad, b0, b1)
    def update_q(self, grad, b0, b1=None): return self.update(CUT_Q, grad, b0, b1)

    @property
    def kappa(self): return getattr(lib(), self._pre + "_kappa")(self.h)
    @property
    def tsq(self): return getattr(lib(), self._pre + "_tsq")(self.h)

    @property
    def mq(self):
        """Writable numpy view of the oracle's n*n matrix."""
        p = getattr(lib(), self._pre + "_mq")(self.h)
        return np.ctypeslib.as_array(p, shape=(self.n, self.n))

    @property
    def xc(self):
        p = getattr(lib(), self._pre + "_xc")(self.h)
        return np.ctypeslib.as_array(p, shape=(self.n,))

    def set_xc(self, x):
        self.xc[:] = np.asarray(x, dtype=np.float64)


class OracleEll(_Space):
    """Ell (src/ell.rs) as restated by the oracle."""
    _pre = "orc_ell"

    def update_rowwise(self, kind, grad, b0, b1=None):
        g = _arr(grad, self.n)
        return lib().orc_ell_update_rowwise(self.h, kind, _ptr(g), float(b0), int(b1 is not None),
                                            0.0 if b1 is None else float(b1))

    def update_rowwise_mt(self, kind, grad, b0, b1=None):
        """NOT the reference's loop: the row-wise form on all OpenMP threads (bit-identical to update_rowwise)."""
        g = _arr(grad, self.n)
        return lib().orc_ell_update_rowwise_mt(self.h, kind, _ptr(g), float(b0), int(b1 is not None),
                                               0.0 if b1 is None else float(b1))

    def set_no_defer_trick(self, flag): lib().orc_ell_set_no_defer_trick(self.h, int(flag))
    def set_use_parallel_cut(self, flag): lib().orc_ell_set_use_parallel_cut(self.h, int(flag))


class OracleEllStable(_Space):
    """EllStable (src/ell_stable.rs) as restated by the oracle (bug-compatible by default)."""
    _pre = "orc_ellstable"

    def set_corrected(self, flag): lib().orc_ellstable_set_corrected(self.h, int(flag))


def set_num_threads(nthreads: int = 0) -> int:
    """OpenMP team of update_rowwise_mt (0: leave as is); returns the team size in force."""
    return int(lib().orc_set_num_threads(int(nthreads)))


def cpu_share() -> int:
    """CPUs this process may really use: the affinity mask, capped by the cgroup's CPU quota when there is one (a GPU
    box shows every logical CPU of the host to a job that owns a 16-CPU share)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def rows_gemv(n, row0, nrows, mq_local, grad, gt_full):
    mq_local = _arr(mq_local, nrows * n)
    grad = _arr(grad, n)
    assert gt_full.dtype == np.float64 and gt_full.size == n and gt_full.flags.c_contiguous
    lib().orc_rows_gemv(n, row0, nrows, _ptr(mq_local), _ptr(grad), _ptr(gt_full))


def lowpass_case(corrected: bool = False):
    """(wpass, wstop, lp_sq, up_sq, sp_sq) of create_lowpass_case (src/oracles/lowpass_oracle.rs:153-167);
    corrected=True: the constants the reference evidently meant (see lowpass_oracle.h)."""
    out = (C.c_double * 5)()
    lib().orc_lowpass_case(int(corrected), out)
    return tuple(out)


class OracleLowpass:
    """LowpassOracle (src/oracles/lowpass_oracle.rs) as restated by the oracle."""

    def __init__(self, ndim, wpass, wstop, lp_sq, up_sq, sp_sq):
        self.n = int(ndim)
        self.p = lib().orc_lowpass_new(self.n, wpass, wstop, lp_sq, up_sq, sp_sq)
        assert self.p

    @classmethod
    def create_case(cls, ndim, corrected=False):
        return cls(ndim, *lowpass_case(corrected))

    def __del__(self):
        p, self.p = getattr(self, "p", None), None
        if p and _lib is not None:
            _lib.orc_lowpass_free(p)

    @property
    def s(self):
        return self.p.contents

    @property
    def spectrum(self):
        return np.ctypeslib.as_array(self.s.spectrum, shape=(int(self.s.mdim), self.n))

    def state(self):
        c = self.s
        return dict(more_alt=c.more_alt, idx1=c.idx1, idx2=c.idx2, idx3=c.idx3, kmax=c.kmax, nwpass=c.nwpass,
                    nwstop=c.nwstop, fmax=c.fmax, sp_sq=c.sp_sq)

    def assess_feas(self, x):
        """None, or (grad, (beta0, beta1 or None))"""
        x = _arr(x, self.n)
        g = np.empty(self.n)
        b0, b1, h = C.c_double(), C.c_double(), C.c_int()
        if not lib().orc_lowpass_assess_feas(self.p, _ptr(x), _ptr(g), C.byref(b0), C.byref(h), C.byref(b1)):
            return None
        return g, (b0.value, b1.value if h.value else None)

    def assess_optim(self, x, gamma):
        """((grad, (beta0, beta1 or None)), shrunk, gamma)"""
        x = _arr(x, self.n)
        g = np.empty(self.n)
        b0, b1, h, sh, gm = C.c_double(), C.c_double(), C.c_int(), C.c_int(), C.c_double(gamma)
        rc = lib().orc_lowpass_assess_optim(self.p, _ptr(x), C.byref(gm), _ptr(g), C.byref(b0), C.byref(h),
                                            C.byref(b1), C.byref(sh))
        if rc < 0:
            raise IndexError("feasible point without a stopband maximum")
        return (g, (b0.value, b1.value if h.value else None)), bool(sh.value), gm.value

    def cutting_plane_optim(self, space, gamma, max_iters, tol):
        """(x_best or None, niter, gamma, last_status) -- src/cutting_plane.rs:286-313"""
        kind = 0 if isinstance(space, OracleEll) else 1
        xb = np.zeros(self.n)
        hb, ls, gm = C.c_int(), C.c_int(), C.c_double(gamma)
        niter = lib().orc_lowpass_cutting_plane_optim(self.p, kind, space.h, C.byref(gm), int(max_iters), float(tol),
                                                      _ptr(xb), C.byref(hb), C.byref(ls))
        return (xb if hb.value else None), int(niter), gm.value, ls.value

    def cutting_plane_feas(self, space, max_iters, tol):
        kind = 0 if isinstance(space, OracleEll) else 1
        xb = np.zeros(self.n)
        hb, ls = C.c_int(), C.c_int()
        niter = lib().orc_lowpass_cutting_plane_feas(self.p, kind, space.h, int(max_iters), float(tol), _ptr(xb),
                                                     C.byref(hb), C.byref(ls))
        return (xb if hb.value else None), int(niter), ls.value


def ell_batch_run(kinds, grads, b0, has_b1, b1, kappa0=1.0, want_state=True):
    """B identity Ell spaces, K cuts each (loop over orc_ell_update).  Arrays [K][B] / [K][B][n]."""
    grads = np.ascontiguousarray(grads, dtype=np.float64)
    K, B, n = grads.shape
    kinds = np.ascontiguousarray(kinds, dtype=np.int32)
    b0 = np.ascontiguousarray(b0, dtype=np.float64)
    has_b1 = np.ascontiguousarray(has_b1, dtype=np.int32)
    b1 = np.ascontiguousarray(b1, dtype=np.float64)
    status = np.empty((K, B), dtype=np.int32)
    mq = np.empty((B, n, n)) if want_state else None
    xc = np.empty((B, n)) if want_state else None
    kap = np.empty(B) if want_state else None
    ok = lib().orc_ell_batch_run(B, n, K, _ptr(kinds), _ptr(grads), _ptr(b0), _ptr(has_b1), _ptr(b1), float(kappa0),
                                 _ptr(status), _ptr(mq), _ptr(xc), _ptr(kap))
    return int(ok), status, mq, xc, kap


class OracleLDLT:
    """LDLTMgr (src/oracles/ldlt_mgr.rs) as restated by the oracle."""

    def __init__(self, ndim, _ptr_=None):
        self.n = int(ndim)
        self._own = _ptr_ is None
        self.p = lib().orc_ldlt_new(self.n) if _ptr_ is None else _ptr_

    def __del__(self):
        p, self.p = getattr(self, "p", None), None
        if p and self._own and _lib is not None:
            _lib.orc_ldlt_free(p)

    @property
    def pos(self):
        return (int(self.p.contents.pos0), int(self.p.contents.pos1))

    @property
    def wit(self):
        return np.ctypeslib.as_array(self.p.contents.wit, shape=(self.n,))

    @property
    def storage(self):
        return np.ctypeslib.as_array(self.p.contents.storage, shape=(self.n, self.n))

    def factorize(self, mat):
        mat = _arr(mat, self.n * self.n)
        return bool(lib().orc_ldlt_factorize(self.p, _ptr(mat)))

    def factor(self, get_elem, allow_semidefinite=False):
        cb = _ELEM_FN(lambda ctx, i, j: float(get_elem(i, j)))
        fn = lib().orc_ldlt_factor_semidefinite if allow_semidefinite else lib().orc_ldlt_factor
        return bool(fn(self.p, cb, None))

    def is_spd(self):
        return self.pos[1] == 0

    def witness(self):
        assert not self.is_spd()
        return lib().orc_ldlt_witness(self.p)

    def sym_quad(self, mat):
        mat = _arr(mat, self.n * self.n)
        return lib().orc_ldlt_sym_quad(self.p, _ptr(mat))

    def sqrt(self):
        assert self.is_spd()
        r = np.empty((self.n, self.n))
        lib().orc_ldlt_sqrt(self.p, _ptr(r))
        return r


class OracleLMI:
    """LMIOracle (mode 0, src/oracles/lmi_oracle.rs) / LMI0Oracle (mode 1, src/oracles/lmi0_oracle.rs)."""

    def __init__(self, mat_f, mat_b=None):
        mat_f = np.ascontiguousarray(mat_f, dtype=np.float64)
        self.n, self.m = int(mat_f.shape[0]), int(mat_f.shape[1])
        assert mat_f.shape == (self.n, self.m, self.m)
        mode = 0 if mat_b is not None else 1
        mb = None if mat_b is None else _arr(mat_b, self.m * self.m)
        self.p = lib().orc_lmi_new(mode, self.n, self.m, _ptr(mat_f), _ptr(mb))

    def __del__(self):
        p, self.p = getattr(self, "p", None), None
        if p and _lib is not None:
            _lib.orc_lmi_free(p)

    @property
    def ldlt(self):
        return OracleLDLT(self.m, _ptr_=self.p.contents.ldlt)

    def assess_feas(self, x):
        """None, or (g, ep)"""
        x = _arr(x, self.n)
        g = np.empty(self.n)
        ep = C.c_double()
        if not lib().orc_lmi_assess_feas(self.p, _ptr(x), _ptr(g), C.byref(ep)):
            return None
        return g, ep.value
