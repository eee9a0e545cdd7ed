"""CPU: libellhip.so loads, exports every entry point include/ellhip.h declares, and refuses to
compute without a HIP device (there is no CPU fallback to fall into)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = set()
    for header in ("ellhip.h", "ellhip_lowpass.h", "ellhip_batch.h", "ellhip_lmi.h", "ellhip_sharded.h"):
        src = open(os.path.join(ROOT, "include", header)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names.update(re.findall(r"\b(ellhip_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_header_declares_what_the_binding_lists():
    import ellalgo_rs_amd as pkg
    assert declared_functions() == sorted(pkg.capi.EXPORTS)


@pytest.mark.parametrize("name", declared_functions())
def test_symbol_exported(name):
    import ellalgo_rs_amd as pkg
    lib = C.CDLL(pkg.capi.lib_path())
    assert getattr(lib, name) is not None


def test_version_string():
    import ellalgo_rs_amd as pkg
    assert pkg.capi.load().ellhip_version().decode().startswith("ellhip ")


def test_no_device_means_loud_failure_not_fallback():
    import ellalgo_rs_amd as pkg
    lib = pkg.capi.load()
    if lib.ellhip_device_count() > 0:
        pytest.skip("a HIP device is visible here")
    h = C.c_void_p()
    rc = lib.ellhip_create(C.byref(h), 0, 4, 1.0, None, None, None, -1)
    assert rc == pkg.capi.E_NODEVICE and not h.value
    assert b"no HIP device" in lib.ellhip_last_error()
    out = (C.c_double * 3)()
    assert lib.ellhip_calc(4, 1, 0, 0.05, 0, 0.0, 0.01, out, -1) == pkg.capi.E_NODEVICE
    with pytest.raises(pkg.capi.EllHipError):
        pkg.Ell.new_with_scalar(1.0, np.zeros(4))
    with pytest.raises(pkg.capi.EllHipError):
        pkg.EllStable.new_with_scalar(1.0, np.zeros(4))
    o = C.c_void_p()
    assert lib.ellhip_lowpass_create(C.byref(o), 8, 0.12, 0.2, 0.9, 1.1, 0.01, None, -1) == pkg.capi.E_NODEVICE
    with pytest.raises(pkg.capi.EllHipError):
        pkg.create_lowpass_case(8)
    assert lib.ellhip_batch_create(C.byref(o), 4, 8, None, None, None, None, -1) == pkg.capi.E_NODEVICE
    with pytest.raises(pkg.capi.EllHipError):
        pkg.EllBatch.new_with_scalar(np.ones(4), np.zeros((4, 8)))
    assert lib.ellhip_lmi_create(C.byref(o), 0, 2, None, (C.c_double * 4)(1, 0, 0, 1), -1) == pkg.capi.E_NODEVICE
    with pytest.raises(pkg.capi.EllHipError):
        pkg.LDLTMgr(3).factorize(np.eye(3))


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under ellalgo-rs_amd/ may reference it."""
    pkg_dir = os.path.join(ROOT, "ellalgo-rs_amd")
    offenders = []
    for base, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".h", ".cpp")):
                text = open(os.path.join(base, f), errors="replace").read()
                if re.search(r"(from|import)\s+oracle\b|ell_oracle\.h|oracle/lowpass_oracle|[\"<]lowpass_oracle\.h[\">]|libell_oracle", text):
                    offenders.append(os.path.join(base, f))
    assert not offenders, offenders


def test_one_hip_runtime_whatever_the_import_order():
    """PyTorch-ROCm bundles its own libamdhip64 / libhsa-runtime64 and opens them by path; the package makes sure
    libellhip.so and torch share ONE runtime per process in either import order (a second runtime on the same GPU
    showed up as a rare hang of the first torch call).  Checked in fresh interpreters through /proc/self/maps."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = r"""
import sys
sys.path.insert(0, %r)
order = sys.argv[1]
if order == "torch-first":
    import torch
import ellalgo_rs_amd as pkg
pkg.capi.load()
if order == "lib-first":
    import torch
libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime64" in l})
print(";".join(libs))
""" % root
    for order in ("lib-first", "torch-first"):
        out = subprocess.run([sys.executable, "-c", prog, order], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr
        libs = out.stdout.strip().split(";")
        assert sum("libamdhip64" in l for l in libs) == 1, (order, libs)
        assert sum("libhsa-runtime64" in l for l in libs) == 1, (order, libs)
