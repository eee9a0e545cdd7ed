"""CPU: the device-free parts of include/ellhip_sharded.h -- the partition function (which must cut the matrix exactly
as the torch.distributed orchestration of ellalgo-rs_amd/sharded.py does, so that a host may mix the two) and the
argument checks of ellhip_sharded_create."""
import ctypes as C

import numpy as np
import pytest


@pytest.mark.parametrize("n", [64 * 5, 64 * 7, 64 * 33, 1024, 4096, 16384, 32768])
def test_partition_matches_the_python_orchestration(n):
    from ellalgo_rs_amd import sharded, sharded_abi
    for world in range(1, 9):
        if n // 64 >= world:
            rows = [sharded_abi.partition(n, world, r, True) for r in range(world)]
            assert rows == [sharded.partition_symmetric(n, world, r) for r in range(world)]
            assert rows[0][0] == 0 and sum(nr for _, nr in rows) == n
            assert all(rows[r][0] + rows[r][1] == rows[r + 1][0] for r in range(world - 1))
            assert all(r0 % 64 == 0 and nr > 0 for r0, nr in rows)
        if n % world == 0:
            assert [sharded_abi.partition(n, world, r, False) for r in range(world)] == \
                   [sharded.partition(n, world, r) for r in range(world)]


def test_symmetric_shards_have_equal_trapezoid_areas():
    from ellalgo_rs_amd import sharded_abi
    n, world = 32768, 8
    areas = []
    for r in range(world):
        r0, nr = sharded_abi.partition(n, world, r, True)
        areas.append(((r0 + nr) ** 2 - r0 ** 2) / 2)
    assert max(areas) / min(areas) < 1.05   # boundaries are rounded to whole 64-row strips


def test_bad_arguments_are_refused():
    import ellalgo_rs_amd as pkg
    lib = pkg.capi.load()
    r0, nr = C.c_int64(), C.c_int64()
    assert lib.ellhip_sharded_partition(1000, 3, 0, 0, C.byref(r0), C.byref(nr)) == pkg.capi.E_INVALID   # 1000 % 3
    assert lib.ellhip_sharded_partition(1000, 2, 0, 1, C.byref(r0), C.byref(nr)) == pkg.capi.E_INVALID   # not k * 64
    assert lib.ellhip_sharded_partition(128, 3, 0, 1, C.byref(r0), C.byref(nr)) == pkg.capi.E_INVALID    # 2 strips, 3 ranks
    assert lib.ellhip_sharded_partition(128, 2, 2, 0, C.byref(r0), C.byref(nr)) == pkg.capi.E_INVALID    # rank out of range
    h = C.c_void_p()
    xc = np.zeros(128)
    p = xc.ctypes.data_as(C.c_void_p)
    # two ranks need a communicator; the symmetric partition needs the recorded schedule; depth must be 1 / 8 / 16
    assert lib.ellhip_sharded_create(C.byref(h), 128, 1.0, None, None, p, -1, 0, 2, None, None, 0, 1) == pkg.capi.E_INVALID
    assert lib.ellhip_sharded_create(C.byref(h), 128, 1.0, None, None, p, -1, 0, 1, None, None, 1, 1) == pkg.capi.E_INVALID
    assert lib.ellhip_sharded_create(C.byref(h), 128, 1.0, None, None, p, -1, 0, 1, None, None, 0, 4) == pkg.capi.E_INVALID
    if lib.ellhip_device_count() == 0:   # no CPU fallback here either
        assert lib.ellhip_sharded_create(C.byref(h), 128, 1.0, None, None, p, -1, 0, 1, None, None, 0, 1) == pkg.capi.E_NODEVICE
        assert not h.value


def _run(code, env):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable, "-c", code], cwd=root, env={**os.environ, **env}, capture_output=True, text=True, timeout=300)


def test_missing_rccl_is_an_error_code_not_a_crash():
    """RCCL absent: ellhip_sharded_unique_id / _create must fail with ELLHIP_E_NORCCL and a message (the first version
    called dlerror() twice and crashed in strlen(NULL)).  Own process: the library opens RCCL once."""
    code = (
        "import ctypes as C\n"
        "import ellalgo_rs_amd as pkg\n"
        "L = pkg.capi.load()\n"
        "buf = (C.c_char * 128)()\n"
        "rc = L.ellhip_sharded_unique_id(buf)\n"
        "msg = L.ellhip_last_error().decode()\n"
        "assert rc == pkg.capi.E_NORCCL, rc\n"
        "assert 'could not be opened' in msg and 'no_such_librccl' in msg, msg\n"
        "rc2 = L.ellhip_sharded_unique_id(buf)\n"
        "assert rc2 == pkg.capi.E_NORCCL\n"
        "print('ok')\n")
    r = _run(code, {"ELLHIP_RCCL_PATH": "/nonexistent/no_such_librccl.so"})
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.returncode, r.stdout, r.stderr)


def test_rccl_path_selects_the_library():
    """ELLHIP_RCCL_PATH names the collective library to open: the in-process stand-in of the multi-rank tests."""
    from cpp_build import build_fake_rccl
    code = (
        "import ctypes as C\n"
        "import ellalgo_rs_amd as pkg\n"
        "L = pkg.capi.load()\n"
        "buf = (C.c_char * 128)()\n"
        "assert L.ellhip_sharded_unique_id(buf) == 0, L.ellhip_last_error()\n"
        "assert bytes(buf[:8]) == b'FAKERCCL'\n"
        "print('ok')\n")
    r = _run(code, {"ELLHIP_RCCL_PATH": build_fake_rccl()})
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.returncode, r.stdout, r.stderr)


def test_custom_collective_needs_both_callbacks():
    import ctypes as C
    import ellalgo_rs_amd as pkg
    L = pkg.capi.load()
    h = C.c_void_p()
    rc = L.ellhip_sharded_create_custom(C.byref(h), 128, 1.0, None, None, None, 0, 0, 2, 0, 1, None, None, None)
    assert rc == pkg.capi.E_INVALID
    # symmetric shards below the lower-triangle schedule's smallest size are refused up front (before any device work)
    rc = L.ellhip_sharded_create(C.byref(h), 192, 1.0, None, None, None, 0, 0, 1, None, None, pkg.capi.SHARD_SYMMETRIC, 8)
    assert rc == pkg.capi.E_INVALID and b"n >= 512" in L.ellhip_last_error()
