"""The public headers are plain C (C99, no C++ needed), and a plain-C program linked against libellhip.so runs the
reference's first known answer through the drop-in boundary."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "c_abi_demo.c")


def test_headers_are_valid_c99():
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", SRC])


@pytest.mark.gpu
def test_plain_c_program_through_the_boundary(gpu):
    out_dir = os.path.join(ROOT, "tests", "cpp", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "c_abi_demo")
    libdir = os.path.join(ROOT, "ellalgo-rs_amd")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-o", exe, SRC, "-L" + libdir, "-lellhip", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lm"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", (r.returncode, r.stdout, r.stderr)
