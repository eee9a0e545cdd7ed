"""GPU: the matrix-core product pass in its queue forms (k_symm_mfma_q, k_symm_mfma_q2: workgroups launched once draw the 64-row tiles from a
counter; 16 / 32 gradients per pass) gives the partial sums of the grid form round 3 shipped (k_symm_mfma, one workgroup per tile, 16
gradients per pass) BIT FOR BIT -- whole matrices, a symmetric row shard, full and ragged groups, many and few workgroups, both segment
widths.  Kernel-level check (tests/cpp/symm_queue_check.hip, built with hipcc against csrc/ell_kernels.hpp): the C ABI no longer reaches
the grid form.  The reference does this product in src/ell.rs:97-103 (`dot_mv`), one gradient at a time."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_queue_forms_of_the_product_pass_equal_the_grid_form_to_the_bit():
    src = os.path.join(ROOT, "tests", "cpp", "symm_queue_check.hip")
    out_dir = os.path.join(ROOT, "tests", "cpp", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "symm_queue_check")
    deps = [src] + [os.path.join(ROOT, "ellalgo-rs_amd", "csrc", f) for f in ("ell_kernels.hpp", "ellcalc_device.hpp")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                               "-I", os.path.join(ROOT, "include"), "-o", exe, src])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    cases = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(cases) == 9, (r.stdout[-2000:], r.stderr[-2000:])
    for c in cases:
        assert c["rowpart_identical"] and c["colpart_identical"] and c["nonzero"], c
        assert c["queue_drawn"] >= c["tiles"], c      # every tile was drawn (plus one miss per workgroup)
    assert r.returncode == 0
