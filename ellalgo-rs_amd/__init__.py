"""ellalgo-rs_amd -- MI355X-native ellipsoid-update engine behind ellalgo-rs's SearchSpace API.

The product is libellhip.so (HIP kernels + the C ABI of include/ellhip.h).  This package holds its
sources (csrc/), the build recipe, the C++ host mirror of the reference's traits and drivers
(host/ellhip/*.hpp) and a thin ctypes mirror used by the tests and the benchmark.
"""
from . import build, capi, synth  # noqa: F401
from .ell import (CutStatus, Ell, EllStable, ParallelCut, SingleCut, calc)  # noqa: F401
from .batch import EllBatch  # noqa: F401
from .lmi import LDLTMgr, LMI0Oracle, LMIOracle  # noqa: F401
from .lowpass import LowpassOracle, create_lowpass_case, lowpass_case_constants  # noqa: F401
from .sharded_abi import ShardedEllAbi  # noqa: F401

__all__ = ["build", "capi", "synth", "CutStatus", "Ell", "EllStable", "ParallelCut", "SingleCut", "calc",
           "EllBatch", "LDLTMgr", "LMIOracle", "LMI0Oracle", "LowpassOracle", "create_lowpass_case", "lowpass_case_constants"]
