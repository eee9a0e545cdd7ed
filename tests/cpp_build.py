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


def build_runner(src: str, backend: str) -> str:
    """backend: 'oracle' (links oracle/libell_oracle.so) or 'hip' (links ellalgo-rs_amd/libellhip.so)."""
    os.makedirs(OUT, exist_ok=True)
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


def run_json_lines(exe, *args, timeout=600):
    import json
    out = subprocess.run([exe, *args], check=True, capture_output=True, text=True, timeout=timeout).stdout
    return {d["case"]: d for d in (json.loads(line) for line in out.splitlines() if line.startswith("{"))}
