"""Build recipe for libellhip.so (hipcc, gfx950 only, in-tree)."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libellhip.so")
REPO_ROOT = os.path.dirname(PKG_DIR)

SOURCES = ["ellhip_capi.hip"]
# every header under csrc/ is a dependency of the one translation unit (tests/test_build_recipe.py checks that each
# `#include "..."` reachable from SOURCES is in this list, so an edited header can never ship a stale libellhip.so)
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp"))
PUBLIC_HEADERS = sorted(f for f in os.listdir(os.path.join(REPO_ROOT, "include")) if f.endswith(".h"))

# -ffp-contract=off: the reference never fuses a*b+c (two roundings per multiply-add); keeping
# that makes the rank-1 pass bit-identical to the CPU arithmetic for the same gt.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libellhip.so")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(REPO_ROOT, "include", f) for f in PUBLIC_HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP engine for gfx950 into ellalgo-rs_amd/libellhip.so."""
    if not force and not needs_build():
        return LIB_PATH
    extra = os.environ.get("ELLHIP_EXTRA_HIPCC_FLAGS", "").split()  # tuning builds only
    cmd = [_hipcc()] + HIPCC_FLAGS + extra + ["-I", os.path.join(REPO_ROOT, "include"), "-o", LIB_PATH] + \
          [os.path.join(CSRC, f) for f in SOURCES] + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


HOST_DIR = os.path.join(PKG_DIR, "host")
LIVE_LOOP_SRC = os.path.join(HOST_DIR, "bench", "live_loop.cpp")
LIVE_LOOP_EXE = os.path.join(HOST_DIR, "bench", "live_loop")


def build_host_tools(force: bool = False, verbose: bool = False) -> str:
    """host/bench/live_loop: the C++ drivers (host/ellhip/*.hpp) around the C ABI with a host oracle -- what bench.py times as
    `live_loop`.  Plain g++: the host side links libellhip.so and nothing else."""
    deps = [LIVE_LOOP_SRC, LIB_PATH] + [os.path.join(HOST_DIR, "ellhip", f) for f in os.listdir(os.path.join(HOST_DIR, "ellhip"))] + \
           [os.path.join(REPO_ROOT, "include", f) for f in PUBLIC_HEADERS]
    if not force and os.path.exists(LIVE_LOOP_EXE) and all(os.path.getmtime(d) <= os.path.getmtime(LIVE_LOOP_EXE) for d in deps):
        return LIVE_LOOP_EXE
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-o", LIVE_LOOP_EXE, LIVE_LOOP_SRC, "-L" + PKG_DIR, "-lellhip",
           "-Wl,-rpath," + PKG_DIR, "-Wl,-rpath,$ORIGIN/../..", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIVE_LOOP_EXE


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_host_tools(force=True, verbose=True))
