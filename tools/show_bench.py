#!/usr/bin/env python3
"""Pretty-print one or more bench.py JSON outputs (development helper)."""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
    except Exception as e:  # noqa: BLE001
        print(path, "unreadable:", e)
        continue
    r = d["roofline"]
    pk = {k: (round(v["avg_ms"], 4), round(v.get("GBps", 0))) for k, v in r.get("per_kernel", {}).items()}
    if "per_kernel" not in r:
        r["whole_update"] = {"frac": r.get("frac", 0)}
        pk = {"kernel_ms": round(r.get("avg_launch_ms", 0), 4), "GBps": round(r.get("achieved", 0))}
    print(f'{d["config"]["workload"]:18s} {d["config"].get("schedule","")[:9]:9s} d{d["config"].get("defer_depth",1)} gpus={d["n_gpus"]} upd/s={d["value"]:9.1f} '
          f'ms={d["ms_per_step"]:.4f} dom={r.get("kernel")} frac={r.get("frac", 0):.3f} whole={(r.get("whole_update") or r.get("whole_iteration"))["frac"]:.3f} {pk}')
    for o in d.get("other_schedules", []):
        pk = {k: (round(v["avg_ms"], 4), round(v.get("GBps", 0))) for k, v in o["per_kernel"].items()}
        print(f'{"":18s} {o["schedule"][:22]:22s} d{o["defer_depth"]}        upd/s={o["updates_per_s"]:9.1f} ms={o["ms_per_step"]:.4f} '
              f'whole={o["whole_update"]["frac"]:.3f} {pk}')
    for o in d.get("other_configs", []):
        ro = o["roofline"]
        pk = {k: (round(v["avg_ms"], 4), round(v.get("GBps", 0))) for k, v in ro.get("per_kernel", {}).items()}
        print(f'  + {o["workload"]:18s} {o["schedule"][:9]:9s} d{o["defer_depth"]} steps={o["steps"]:4d} upd/s={o["updates_per_s"]:9.1f} ms={o["ms_per_step"]:.4f} '
              f'dom={ro.get("kernel")} frac={ro.get("frac", 0):.3f} whole={ro["whole_update"]["frac"]:.3f} {pk}')
    if "host_call_path" in d:
        hp = d["host_call_path"]
        print(f'{"":18s} host-call path ' + " ".join(f"{k}={v:.1f}" for k, v in hp.items() if k.startswith("updates_per_s")), end="")
    if "cpu_baseline" in d:
        print(f'   cpu {d["cpu_baseline"]["value"]:.3f} /s', end="")
        if "gpu_same_sample_iterations_per_s" in d["cpu_baseline"]:
            print(f'   (device on the same sample {d["cpu_baseline"]["gpu_same_sample_iterations_per_s"]:.1f} /s)', end="")
    print()
