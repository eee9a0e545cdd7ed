#!/usr/bin/env python3
"""Split a rocprofv3 kernel trace by (kernel, grid size): the default bench command runs every BASELINE configuration in
one process, so `--stats` averages e.g. k_symv over the n = 16384 and n = 32768 launches together; this prints the
per-grid averages (the first `skip` launches of each group, warm-up / first-touch, are reported but not averaged when the
group has more than 4 * skip launches).

usage: trace_by_grid.py <dir with *_kernel_trace.csv> [out.csv]"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
groups = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "ellhip::" not in name:
            continue
        short = name.split("ellhip::")[1].split("(")[0]
        grid = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        groups[(short, grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
w = csv.writer(out)
w.writerow(["kernel", "workgroups_x", "grid_y", "grid_z", "launches", "avg_ns", "median_ns", "min_ns", "max_ns", "total_ns"])
for (short, grid), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
    s = sorted(v)
    w.writerow([short, grid[0], grid[1], grid[2], len(v), round(sum(v) / len(v), 1), s[len(s) // 2], s[0], s[-1], sum(v)])
