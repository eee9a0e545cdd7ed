import os
import sys

import pytest

# The oracle's row-parallel variant (OpenMP) is used by a few full-size tests.  A GPU box exposes every logical CPU of
# the host (256) while a job may run on a 16-CPU share: cap the team before libgomp is loaded (256 threads spinning on
# 16 CPUs made one update take ~0.5 s instead of ~0.1 s).
_aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(_aff, 16))))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_count():
    try:
        import ellalgo_rs_amd as pkg
        return pkg.capi.load().ellhip_device_count()
    except Exception:
        return 0


@pytest.fixture(scope="session")
def gpu():
    """The product package, with a hard requirement that a HIP device and the HIP library exist."""
    import ellalgo_rs_amd as pkg
    lib = pkg.capi.load()  # raises if libellhip.so is missing -- no fallback
    if lib.ellhip_device_count() <= 0:
        pytest.fail("test is marked gpu but no HIP device is visible")
    return pkg


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    oracle.lib()
    return oracle


FACTORY_DEFAULTS = {"AUTO_DEFER": 1, "SYMV": 1, "SYMV_MIN_N": 5120, "APPLY_LOWER": 1, "APPLY_KERNEL": -1, "FUSE_DOTS": 1,
                    "STABLE_SOLVE": 3, "STABLE_FACTOR": 2, "PAD": -1, "LP_GRID": 0, "LP_WIDE": -1, "BATCH_THREADS": 0, "RESIDENT": 1, "OVERLAP": 1, "LOOKAHEAD": 32, "QUEUE_DEPTH": 48, "STAGE_DIRECT": 1}


@pytest.fixture(autouse=True)
def _factory_default_options(request):
    """Tests pin kernel forms with ellhip_set_default_option (tests/util.py: set_default); every test starts from and
    leaves behind the library's factory defaults."""
    yield
    if "ellalgo_rs_amd" in sys.modules:
        capi = sys.modules["ellalgo_rs_amd"].capi
        if capi._lib is not None:
            for name, v in FACTORY_DEFAULTS.items():
                capi.set_default_option(getattr(capi, "OPT_" + name), v)
