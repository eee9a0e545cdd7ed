#!/usr/bin/env python3
"""Summarise rocprofv3 output dirs into profiles/<round>/: kernel stats CSV + PMC traffic JSON.

usage: pmc_summary.py <stats_dir> <fetch_dir> <write_dir> <out_dir> <tag> <workload>
HBM bytes per launch = 2*FETCH_SIZE*1024 (gfx950 half-count correction for 16 B/lane streams,
MI355X_MICROARCH.md, HBM section) + WRITE_SIZE*1024; one PMC pass per counter."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

stats_dir, fetch_dir, write_dir, out_dir, tag, workload = sys.argv[1:7]
os.makedirs(out_dir, exist_ok=True)
for f in glob.glob(os.path.join(stats_dir, "**", "*_kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(out_dir, f"{tag}_kernel_stats.csv"))
vals = collections.defaultdict(dict)
for name, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == name:
                agg[row["Kernel_Name"]].append(float(row["Counter_Value"]))
        for k, v in agg.items():
            if "ellhip::k_" in k and "fill" not in k:
                short = k.split("ellhip::")[1].split("(")[0]
                vals[short][name] = sum(v) / len(v)
                vals[short]["launches_" + name] = len(v)
out = {"workload": workload,
       "units": "FETCH_SIZE/WRITE_SIZE in KiB as reported by rocprofv3; hbm_read_bytes = 2*FETCH_SIZE*1024 "
                "(gfx950 correction), hbm_write_bytes = WRITE_SIZE*1024; separate --pmc passes",
       "kernels": {}}
for k, v in sorted(vals.items()):
    rd = 2 * v.get("FETCH_SIZE", 0) * 1024
    wr = v.get("WRITE_SIZE", 0) * 1024
    out["kernels"][k] = {"FETCH_SIZE_KiB_avg": v.get("FETCH_SIZE"), "WRITE_SIZE_KiB_avg": v.get("WRITE_SIZE"),
                         "launches": v.get("launches_FETCH_SIZE"), "hbm_read_bytes": rd, "hbm_write_bytes": wr,
                         "hbm_bytes_per_launch": rd + wr}
json.dump(out, open(os.path.join(out_dir, f"{tag}_pmc.json"), "w"), indent=1)
for k, v in out["kernels"].items():
    print(f"{k:60s} {v['hbm_bytes_per_launch'] / 1e9:8.4f} GB/launch  (read {v['hbm_read_bytes'] / 1e9:.4f} write {v['hbm_write_bytes'] / 1e9:.4f})")
