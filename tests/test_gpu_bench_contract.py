"""GPU: bench.py honours its output contract -- exactly ONE JSON line on stdout with the driver's fields, the
`roofline` object of the dominant kernel and the `cpu_baseline` object -- for the headline workload and for the
secondary workloads."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

COMMON = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
          "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict, "roofline": dict}


def run_bench(*args, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"stdout must hold exactly one line, got {len(lines)}"
    return json.loads(lines[0])


def check_common(d, steps, warmup):
    for k, t in COMMON.items():
        assert k in d and isinstance(d[k], t), (k, d.get(k))
    assert d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in d["config"]
    r = d["roofline"]
    # (the matrix-core product pass is priced against the FP64 matrix pipe when its window's passes carry more than 20 gradients
    # on average -- above the card's ridge of 9.8 flop/B -- and against HBM otherwise; every other kernel against HBM)
    assert (r["bound"], r["unit"], r["peak"]) in (("hbm", "GB/s", 8000.0), ("mfma", "TFLOP/s", 78.6))
    if r["bound"] == "mfma":
        assert r["kernel"] == "k_symm_mfma" and r["matrix_pipe"]["gradients_per_pass"] > 19.65 and 0.0 < r["hbm_view"]["frac"] < 1.0
    assert r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r


def test_headline_workload_contract():
    d = run_bench("--steps", "16", "--warmup", "8", "--compare-steps", "0", "--profile-steps", "8", "--host-path-steps", "4",
                  "--cpu-budget", "2", "--live-loop-steps", "30")
    check_common(d, 16, 8)
    assert d["metric"].startswith("ellipsoid updates/sec at n=16384") and d["unit"] == "updates/s"
    assert d["config"]["workload"] == "n16384-parallel" and d["scaling"] == "strong"
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["kernel"] == "k_symm_mfma" and r["traffic"] and 0.2 < r["frac"] < 1.0 and d["config"]["lookahead"] == 32 and d["config"]["queue_depth"] == 48
    assert abs(r["alg_bytes_per_launch"] - 4.0 * 16384 ** 2) < 1.0     # lower triangle: 4 n^2 bytes
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "updates/s" and c["value"] > 0 and c["sample"]
    assert c["all_cores"]["threads"] >= 1 and "NOT the reference" in c["all_cores"]["note"]
    assert d["host_call_path"]["updates_per_s"] > 0
    # the live loop (C++ drivers over the C ABI, host oracle) and what a SearchSpace caller can reach, in the objects a record
    # keeper keeps: the headline needs 16 future gradients per pass, the live figures none
    ll = d["live_loop"]
    assert ll["steps"] == 30 and ll["rc"] == 0 and ll["plain_niter"] == ll["pipelined_niter"] == 78
    assert ll["plain_iterations_per_s"] > 1000 and ll["pipelined_iterations_per_s"] > 1000 and ll["defer_depth"] == 24
    cfg = d["config"]
    assert cfg["value_requires_future_gradients"] == 32 and cfg["live_loop_updates_per_s"] == ll["plain_iterations_per_s"]
    assert cfg["host_call_updates_per_s"] == d["host_call_path"]["updates_per_s_depth24"]
    assert cfg["live_loop_updates_per_s"] < d["value"]
    assert cfg["steady_state_updates_per_s"] > 0.8 * d["value"]      # (192 more steps of the timed schedule beside a 16-step region)
    assert r["traffic_measured_in_this_run"] is False and r["per_kernel"]["symv"]["avg_ms"] > 0
    # the as-run figure (the next group's products overlap this group's stage) and the kernel by itself
    assert r["isolated"]["kernel"] == "k_symm_mfma" and 0.3 < r["isolated"]["frac"] < 1.0
    assert r["per_kernel_isolated"]["symv"]["avg_ms"] > 0
    # 16 timed steps, lookahead 32: ONE pass over Q (a group of 16) and ONE apply pass (the flush at the end):
    # (1 * 4 n^2 + 1 * 8 n^2) / 16 = 0.75 n^2 per update
    assert abs(r["whole_update"]["alg_bytes_per_gpu"] - 0.75 * 16384 ** 2) < 1.0
    # the default invocation carries BASELINE.json's other configurations in the same line
    oc = {o["workload"]: o for o in d["other_configs"]}
    assert set(oc) == {"n4096-deep", "n32768-deep", "n16384-ellstable"}
    for wl, o in oc.items():
        assert o["updates_per_s"] > 0 and abs(o["updates_per_s"] - 1e3 / o["ms_per_step"]) < 1e-6 * o["updates_per_s"]
        ro = o["roofline"]
        assert 0.0 < ro["frac"] < 1.0 and ro["kernel"] and 0.0 < ro["whole_update"]["frac"] < 1.0
    assert oc["n32768-deep"]["defer_depth"] == 24 and oc["n4096-deep"]["defer_depth"] == 8
    # 96 steps, lookahead 32, up to 48 recorded inside a run: groups 32 16 | 32 16 and two apply passes,
    # (4 * 4 + 2 * 8) / 96 = 0.33333
    assert abs(oc["n32768-deep"]["roofline"]["whole_update"]["alg_bytes"] - (32.0 / 96.0) * 32768 ** 2) < 4.0


@pytest.mark.parametrize("workload,args", [("n4096-deep", ("--steps", "40", "--warmup", "8", "--compare-steps", "0")),
                                           ("n4096-ellstable", ("--steps", "20", "--warmup", "4")),
                                           ("lowpass-n1024", ("--steps", "200", "--warmup", "50", "--profile-steps", "50")),
                                           ("batch-n16", ("--steps", "10", "--warmup", "2")),
                                           ("lmi-m1024-n128", ("--steps", "3", "--warmup", "1"))])
def test_other_workloads_contract(workload, args):
    d = run_bench("--workload", workload, "--cpu-budget", "1", "--host-path-steps", "0", *args)
    check_common(d, int(args[1]), int(args[3]))
    assert d["config"]["workload"] == workload
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
