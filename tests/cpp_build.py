"""Builds the C++ test runners under tests/cpp/_build (g++; the HIP backend only LINKS libellhip.so)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
OUT = os.path.join(CPP, "_build")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


HIP_HOST_FLAGS = ["-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"]   # <hip/hip_runtime_api.h> from plain g++


def build_fake_rccl() -> str:
    """tests/cpp/fake_rccl.cpp -> _build/libfakerccl.so: the in-process stand-in for librccl the multi-rank tests hand to
    libellhip.so through ELLHIP_RCCL_PATH."""
    os.makedirs(OUT, exist_ok=True)
    lib = os.path.join(OUT, "libfakerccl.so")
    deps = [os.path.join(CPP, "fake_rccl.cpp"), os.path.join(CPP, "inproc_collective.hpp")]
    if _newer(lib, deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", *HIP_HOST_FLAGS, "-o", lib, deps[0],
                               "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-lpthread"])
    return lib


def build_runner(src: str, backend: str) -> str:
    """backend: 'oracle' (links oracle/libell_oracle.so), 'hip' (links ellalgo-rs_amd/libellhip.so) or 'hip+oracle' (both:
    a runner that drives the engine and checks it against the oracle itself)."""
    os.makedirs(OUT, exist_ok=True)
    if backend == "hip+oracle":
        from oracle import oracle
        oracle.build()
        exe = os.path.join(OUT, f"{os.path.splitext(src)[0]}_hip_oracle")
        deps = [os.path.join(CPP, f) for f in os.listdir(CPP) if f.endswith((".cpp", ".hpp"))]
        deps += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
        deps += [os.path.join(ROOT, "ellalgo-rs_amd", "libellhip.so"), os.path.join(ROOT, "oracle", "libell_oracle.so")]
        if _newer(exe, deps):
            hipdir, orcdir = os.path.join(ROOT, "ellalgo-rs_amd"), os.path.join(ROOT, "oracle")
            subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", *HIP_HOST_FLAGS, "-o", exe, os.path.join(CPP, src),
                                   "-L" + hipdir, "-lellhip", "-Wl,-rpath," + hipdir, "-L" + orcdir, "-lell_oracle",
                                   "-Wl,-rpath," + orcdir, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-lpthread",
                                   "-lm"])
        return exe
    exe = os.path.join(OUT, f"{os.path.splitext(src)[0]}_{backend}")
    host = os.path.join(ROOT, "ellalgo-rs_amd", "host", "ellhip")
    deps = [os.path.join(CPP, f) for f in os.listdir(CPP) if f.endswith((".cpp", ".hpp"))]
    deps += [os.path.join(host, f) for f in os.listdir(host)]
    deps += [os.path.join(ROOT, "include", f) for f in ("ellhip.h", "ellhip_lowpass.h", "ellhip_batch.h", "ellhip_lmi.h", "ellhip_sharded.h")]
    if backend == "oracle":
        from oracle import oracle
        oracle.build()
        libdir, lib, define = os.path.join(ROOT, "oracle"), "ell_oracle", "-DBACKEND_ORACLE"
        deps.append(os.path.join(libdir, "libell_oracle.so"))
    else:
        libdir, lib, define = os.path.join(ROOT, "ellalgo-rs_amd"), "ellhip", "-DBACKEND_HIP"
        deps.append(os.path.join(libdir, "libellhip.so"))
    if _newer(exe, deps):
        cmd = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", define, "-o", exe, os.path.join(CPP, src),
               "-L" + libdir, "-l" + lib, "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lm"]
        subprocess.check_call(cmd)
    return exe


def run_json_lines(exe, *args, timeout=600, env=None):
    import json
    full_env = None if env is None else {**os.environ, **env}
    out = subprocess.run([exe, *args], check=True, capture_output=True, text=True, timeout=timeout, env=full_env).stdout
    return {d["case"]: d for d in (json.loads(line) for line in out.splitlines() if line.startswith("{"))}
