"""ctypes binding of include/ellhip.h (libellhip.so).

Loading fails loudly: there is no Python or CPU implementation of the update behind this module.
"""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

SUCCESS, NOSOLN, NOEFFECT, UNKNOWN = 0, 1, 2, 3
CUT_BIAS, CUT_CENTRAL, CUT_Q = 0, 1, 2
SPACE_ELL, SPACE_ELL_STABLE = 0, 1
E_NORCCL = -6
SHARD_EQUAL_BLOCKS, SHARD_SYMMETRIC = 0, 1
NCCL_ID_BYTES = 128
E_INVALID, E_HIP, E_NODEVICE, E_NOMEM, E_STATE = -1, -2, -3, -4, -5
NKERNEL_CLASSES = 14
# option keys (include/ellhip.h, "options")
(OPT_AUTO_DEFER, OPT_SYMV, OPT_SYMV_MIN_N, OPT_APPLY_LOWER, OPT_APPLY_KERNEL, OPT_FUSE_DOTS, OPT_STABLE_SOLVE,
 OPT_STABLE_FACTOR, OPT_PAD, OPT_LP_GRID, OPT_LP_WIDE, OPT_BATCH_THREADS, OPT_RESIDENT, OPT_OVERLAP,
 OPT_LOOKAHEAD, OPT_QUEUE_DEPTH, OPT_RESIDENT_FAULT, OPT_RESIDENT_ABANDONED, OPT_STABLE_MIRRORED, OPT_STAGE_DIRECT) = range(1, 21)
KERNEL_CLASS_NAMES = ("gemv", "scalar", "rank1", "stable_fwd", "stable_bwd", "stable_factor", "fused", "apply",
                      "apply_gemv", "symv", "symv_reduce", "lp_scan", "lp_final", "resident")

# every symbol include/ellhip.h declares
EXPORTS = [
    "ellhip_create", "ellhip_create_shard", "ellhip_clone", "ellhip_destroy", "ellhip_update", "ellhip_tsq",
    "ellhip_get_xc", "ellhip_set_xc", "ellhip_kappa", "ellhip_ndim", "ellhip_get_mq",
    "ellhip_set_no_defer_trick", "ellhip_set_use_parallel_cut", "ellhip_set_defer_depth", "ellhip_defer_depth", "ellhip_flush", "ellhip_queue_primed", "ellhip_set_shard_symmetric", "ellhip_calc", "ellhip_update_begin",
    "ellhip_update_end", "ellhip_gt_dev", "ellhip_set_gt_dev", "ellhip_prime", "ellhip_cut", "ellhip_commit",
    "ellhip_queue_upload", "ellhip_queue_run", "ellhip_queue_run_fused", "ellhip_queue_begin", "ellhip_queue_end",
    "ellhip_queue_prime", "ellhip_queue_cut", "ellhip_queue_commit", "ellhip_queue_results", "ellhip_set_stream",
    "ellhip_synchronize", "ellhip_profile_enable", "ellhip_profile_read", "ellhip_device_count",
    "ellhip_last_error", "ellhip_version",
    "ellhip_set_option", "ellhip_get_option", "ellhip_set_default_option", "ellhip_default_option",
    # include/ellhip_lowpass.h
    "ellhip_lowpass_create", "ellhip_lowpass_destroy", "ellhip_lowpass_assess_feas", "ellhip_lowpass_assess_optim",
    "ellhip_lowpass_state", "ellhip_lowpass_rows_visited", "ellhip_lowpass_get_spectrum", "ellhip_lowpass_optim", "ellhip_lowpass_feas",
    # include/ellhip_batch.h
    "ellhip_batch_create", "ellhip_batch_from_space", "ellhip_batch_destroy", "ellhip_batch_update",
    "ellhip_batch_update_dev", "ellhip_batch_synchronize", "ellhip_batch_stream", "ellhip_batch_get_xc",
    "ellhip_batch_set_xc", "ellhip_batch_get_mq", "ellhip_batch_get_kappa", "ellhip_batch_get_tsq",
    "ellhip_batch_size", "ellhip_batch_ndim", "ellhip_batch_set_no_defer_trick", "ellhip_batch_set_use_parallel_cut",
    # include/ellhip_lmi.h
    "ellhip_lmi_create", "ellhip_lmi_destroy", "ellhip_lmi_assess_feas", "ellhip_lmi_pos", "ellhip_lmi_get_witness",
    "ellhip_lmi_get_storage", "ellhip_lmi_sqrt",
    # include/ellhip_sharded.h
    "ellhip_sharded_partition", "ellhip_sharded_unique_id", "ellhip_sharded_create", "ellhip_sharded_destroy",
    "ellhip_sharded_update", "ellhip_sharded_tsq", "ellhip_sharded_kappa", "ellhip_sharded_get_xc",
    "ellhip_sharded_set_xc", "ellhip_sharded_get_mq_rows", "ellhip_sharded_set_defer_depth", "ellhip_sharded_flush",
    "ellhip_sharded_queue_upload", "ellhip_sharded_queue_run", "ellhip_sharded_queue_run_fused",
    "ellhip_sharded_queue_results", "ellhip_sharded_synchronize", "ellhip_sharded_local", "ellhip_shards_exchange",
    "ellhip_sharded_create_custom", "ellhip_sharded_set_collective",
]


class EllHipError(RuntimeError):
    pass


_lib = None


def lib_path() -> str:
    return _build.LIB_PATH


def _one_hip_runtime_per_process() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so (same SONAMEs as /opt/rocm's) and
    open them BY PATH.  If libellhip.so came first, it has already pulled in /opt/rocm's copies and a later
    `import torch` (the multi-GPU path imports it for RCCL) adds a SECOND HIP + HSA runtime to the process -- two
    runtimes driving one GPU, which showed up as a rare hang of the first torch call.  So in a Python process that
    has PyTorch-ROCm installed, its runtime is opened first and libellhip.so binds to it (SONAME match), whichever
    of the two is imported first.  (C / C++ / Rust hosts link /opt/rocm's runtime and never see torch.)
    ELLHIP_SYSTEM_HIP=1 skips this."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("ELLHIP_SYSTEM_HIP", "0") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    hip = os.path.join(libdir, "libamdhip64.so")
    if os.path.exists(hip):
        try:
            C.CDLL(hip, mode=C.RTLD_GLOBAL)  # its RPATH ($ORIGIN) brings the bundled libhsa-runtime64.so along
        except OSError:
            pass  # not loadable here: fall back to the system runtime


def mapped_runtimes() -> dict:
    """{'libamdhip64': [paths], 'libhsa-runtime64': [paths]} of this process, from /proc/self/maps."""
    found = {"libamdhip64": set(), "libhsa-runtime64": set()}
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                for key in found:
                    if key + ".so" in line:
                        found[key].add(os.path.realpath(line.split()[-1]))
    except OSError:
        pass
    return {k: sorted(v) for k, v in found.items()}


def _assert_one_runtime_mapped(before: dict) -> None:
    """Fail loudly if loading libellhip.so ADDED a second HIP or HSA runtime to the process (see
    _one_hip_runtime_per_process): the binding to PyTorch-ROCm's bundled runtime rests on a SONAME match, and two
    runtimes driving one GPU is the state that hung.  `before`: mapped_runtimes() right before the CDLL.  Copies that
    were already there are not this loader's doing -- `rocprofv3 -- python3 bench.py` preloads /opt/rocm's HSA runtime
    into a process whose `import torch` then maps PyTorch's own -- and are reported once on stderr, not raised.
    ELLHIP_SYSTEM_HIP=1 (the caller chose /opt/rocm's runtime knowingly) skips the check."""
    if os.environ.get("ELLHIP_SYSTEM_HIP", "0") == "1":
        return
    after = mapped_runtimes()
    ours = {k: v for k, v in after.items() if len(v) > 1 and v != before.get(k)}
    if ours:
        raise EllHipError(f"loading libellhip.so mapped a second copy of a GPU runtime: {ours} (before: {before}); import "
                          "order or a SONAME mismatch defeated the single-runtime rule (ellalgo-rs_amd/capi.py)")
    theirs = {k: v for k, v in after.items() if len(v) > 1}
    if theirs:
        import sys
        print(f"[ellhip] note: this process already had two copies of a GPU runtime mapped before libellhip.so was "
              f"loaded (a profiler's preload?): {theirs}", file=sys.stderr)


def load():
    """dlopen libellhip.so (built in-tree by ellalgo-rs_amd/build.py) and type its entry points."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise EllHipError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(the ellipsoid engine has no CPU fallback)")
    _one_hip_runtime_per_process()
    before = mapped_runtimes()
    L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    _assert_one_runtime_mapped(before)
    vp, dbl, i32, i64 = C.c_void_p, C.c_double, C.c_int, C.c_int64
    sig = {
        "ellhip_create": (i32, [C.POINTER(vp), i32, i64, dbl, vp, vp, vp, i32]),
        "ellhip_create_shard": (i32, [C.POINTER(vp), i64, i64, i64, dbl, vp, vp, vp, i32]),
        "ellhip_clone": (i32, [vp, C.POINTER(vp)]),
        "ellhip_destroy": (None, [vp]),
        "ellhip_update": (i32, [vp, i32, vp, dbl, i32, dbl]),
        "ellhip_tsq": (dbl, [vp]),
        "ellhip_get_xc": (i32, [vp, vp]),
        "ellhip_set_xc": (i32, [vp, vp]),
        "ellhip_kappa": (dbl, [vp]),
        "ellhip_ndim": (i64, [vp]),
        "ellhip_get_mq": (i32, [vp, vp]),
        "ellhip_set_no_defer_trick": (i32, [vp, i32]),
        "ellhip_set_use_parallel_cut": (i32, [vp, i32]),
        "ellhip_set_defer_depth": (i32, [vp, i32]),
        "ellhip_queue_primed": (i64, [vp]),
        "ellhip_defer_depth": (i32, [vp]),
        "ellhip_flush": (i32, [vp]),
        "ellhip_set_shard_symmetric": (i32, [vp, i32]),
        "ellhip_calc": (i32, [i64, i32, i32, dbl, i32, dbl, dbl, vp, i32]),
        "ellhip_update_begin": (i32, [vp, i32, vp, dbl, i32, dbl]),
        "ellhip_update_end": (i32, [vp]),
        "ellhip_gt_dev": (vp, [vp]),
        "ellhip_set_gt_dev": (i32, [vp, vp, vp]),
        "ellhip_prime": (i32, [vp, vp]),
        "ellhip_cut": (i32, [vp, i32, dbl, i32, dbl]),
        "ellhip_commit": (i32, [vp, vp]),
        "ellhip_queue_run_fused": (i32, [vp, i64, i64]),
        "ellhip_queue_prime": (i32, [vp, i64]),
        "ellhip_queue_cut": (i32, [vp, i64]),
        "ellhip_queue_commit": (i32, [vp, i64, i64]),
        "ellhip_queue_upload": (i32, [vp, i64, vp, vp, vp, vp, vp]),
        "ellhip_queue_run": (i32, [vp, i64, i64]),
        "ellhip_queue_begin": (i32, [vp, i64]),
        "ellhip_queue_end": (i32, [vp, i64]),
        "ellhip_queue_results": (i32, [vp, vp, vp]),
        "ellhip_set_stream": (i32, [vp, vp]),
        "ellhip_synchronize": (i32, [vp]),
        "ellhip_profile_enable": (i32, [vp, i32]),
        "ellhip_profile_read": (i32, [vp, vp, vp]),
        "ellhip_device_count": (i32, []),
        "ellhip_last_error": (C.c_char_p, []),
        "ellhip_version": (C.c_char_p, []),
        "ellhip_set_option": (i32, [vp, i32, i64]),
        "ellhip_get_option": (i32, [vp, i32, C.POINTER(i64)]),
        "ellhip_set_default_option": (i32, [i32, i64]),
        "ellhip_default_option": (i32, [i32, C.POINTER(i64)]),
        "ellhip_lowpass_create": (i32, [C.POINTER(vp), i64, dbl, dbl, dbl, dbl, dbl, vp, i32]),
        "ellhip_lowpass_destroy": (None, [vp]),
        "ellhip_lowpass_assess_feas": (i32, [vp, vp, vp, C.POINTER(dbl), C.POINTER(i32), C.POINTER(dbl)]),
        "ellhip_lowpass_assess_optim": (i32, [vp, vp, C.POINTER(dbl), vp, C.POINTER(dbl), C.POINTER(i32),
                                              C.POINTER(dbl), C.POINTER(i32)]),
        "ellhip_lowpass_state": (i32, [vp, vp, vp]),
        "ellhip_lowpass_rows_visited": (i64, [vp, i32]),
        "ellhip_lowpass_get_spectrum": (i32, [vp, vp]),
        "ellhip_lowpass_optim": (i32, [vp, vp, C.POINTER(dbl), i64, dbl, vp, C.POINTER(i32), C.POINTER(i64)]),
        "ellhip_lowpass_feas": (i32, [vp, vp, i64, dbl, vp, C.POINTER(i32), C.POINTER(i64)]),
        "ellhip_batch_create": (i32, [C.POINTER(vp), i64, i64, vp, vp, vp, vp, i32]),
        "ellhip_batch_from_space": (i32, [C.POINTER(vp), vp, i64]),
        "ellhip_batch_destroy": (None, [vp]),
        "ellhip_batch_update": (i32, [vp, i64, vp, vp, vp, vp, vp, vp, vp]),
        "ellhip_batch_update_dev": (i32, [vp, i64, vp, vp, vp, vp, vp, vp, vp]),
        "ellhip_batch_synchronize": (i32, [vp]),
        "ellhip_batch_stream": (vp, [vp]),
        "ellhip_batch_get_xc": (i32, [vp, vp]),
        "ellhip_batch_set_xc": (i32, [vp, vp]),
        "ellhip_batch_get_mq": (i32, [vp, vp]),
        "ellhip_batch_get_kappa": (i32, [vp, vp]),
        "ellhip_batch_get_tsq": (i32, [vp, vp]),
        "ellhip_batch_size": (i64, [vp]),
        "ellhip_batch_ndim": (i64, [vp]),
        "ellhip_batch_set_no_defer_trick": (i32, [vp, i32]),
        "ellhip_batch_set_use_parallel_cut": (i32, [vp, i32]),
        "ellhip_lmi_create": (i32, [C.POINTER(vp), i64, i64, vp, vp, i32]),
        "ellhip_lmi_destroy": (None, [vp]),
        "ellhip_lmi_assess_feas": (i32, [vp, vp, vp, C.POINTER(dbl)]),
        "ellhip_lmi_pos": (i32, [vp, vp]),
        "ellhip_lmi_get_witness": (i32, [vp, vp]),
        "ellhip_lmi_get_storage": (i32, [vp, vp]),
        "ellhip_lmi_sqrt": (i32, [vp, vp]),
        # include/ellhip_sharded.h
        "ellhip_sharded_partition": (i32, [i64, i32, i32, i32, C.POINTER(i64), C.POINTER(i64)]),
        "ellhip_sharded_unique_id": (i32, [vp]),
        "ellhip_sharded_create": (i32, [C.POINTER(vp), i64, dbl, vp, vp, vp, i32, i32, i32, vp, vp, i32, i32]),
        "ellhip_sharded_destroy": (None, [vp]),
        "ellhip_sharded_update": (i32, [vp, i32, vp, dbl, i32, dbl]),
        "ellhip_sharded_tsq": (dbl, [vp]),
        "ellhip_sharded_kappa": (dbl, [vp]),
        "ellhip_sharded_get_xc": (i32, [vp, vp]),
        "ellhip_sharded_set_xc": (i32, [vp, vp]),
        "ellhip_sharded_get_mq_rows": (i32, [vp, vp]),
        "ellhip_sharded_set_defer_depth": (i32, [vp, i32]),
        "ellhip_sharded_flush": (i32, [vp]),
        "ellhip_sharded_queue_upload": (i32, [vp, i64, vp, vp, vp, vp, vp]),
        "ellhip_sharded_queue_run": (i32, [vp, i64, i64]),
        "ellhip_sharded_queue_run_fused": (i32, [vp, i64, i64]),
        "ellhip_sharded_queue_results": (i32, [vp, vp, vp]),
        "ellhip_sharded_synchronize": (i32, [vp]),
        "ellhip_sharded_local": (vp, [vp]),
        "ellhip_shards_exchange": (i32, [vp, i32]),
        "ellhip_sharded_create_custom": (i32, [C.POINTER(vp), i64, dbl, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]),
        "ellhip_sharded_set_collective": (i32, [vp, vp, vp, vp]),
    }
    for name in EXPORTS:
        fn = getattr(L, name)  # AttributeError if the library does not export it
        fn.restype, fn.argtypes = sig[name]
    _lib = L
    return L


def check(rc: int, what: str = "") -> int:
    """Raise on a library failure (negative code); pass CutStatus / 0 through."""
    if rc < 0:
        msg = load().ellhip_last_error().decode(errors="replace")
        raise EllHipError(f"{what or 'ellhip'} failed with code {rc}: {msg}")
    return rc


def set_default_option(key: int, value: int) -> None:
    """ellhip_set_default_option: what handles created LATER in this process start with."""
    check(load().ellhip_set_default_option(int(key), int(value)), "ellhip_set_default_option")


def default_option(key: int) -> int:
    v = C.c_int64()
    check(load().ellhip_default_option(int(key), C.byref(v)), "ellhip_default_option")
    return int(v.value)


class default_options:
    """Context manager: `with capi.default_options({capi.OPT_SYMV_MIN_N: 512}): ...` -- handles created inside start with
    these values; the previous defaults come back on exit."""

    def __init__(self, opts):
        self.opts = dict(opts)

    def __enter__(self):
        self.old = {k: default_option(k) for k in self.opts}
        for k, v in self.opts.items():
            set_default_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_default_option(k, v)
        return False
