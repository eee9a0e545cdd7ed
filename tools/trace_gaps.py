#!/usr/bin/env python3
"""From a rocprofv3 kernel trace: for one kernel (default k_symv) the period between consecutive launches, the gap from the
end of one to the start of the next, and what ran inside that gap; medians over the launches of the most common grid.

usage: trace_gaps.py <dir with *_kernel_trace.csv> [kernel substring]"""
import collections
import csv
import glob
import os
import statistics
import sys

d = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "k_symv<"
rows = []
for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        short = name.split("ellhip::")[1].split("(")[0] if "ellhip::" in name else name[:40]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]),
                     r["Queue_Id"]))
rows.sort()
sel = [r for r in rows if key in r[2]]
grid = collections.Counter(r[3] for r in sel).most_common(1)[0][0]
sel = [r for r in sel if r[3] == grid]
per, gap, dur = [], [], []
for a, b in zip(sel, sel[1:]):
    if b[0] - a[0] > 3 * (a[1] - a[0]):
        continue  # another phase of the run in between
    per.append(b[0] - a[0])
    gap.append(b[0] - a[1])
    dur.append(a[1] - a[0])
q = lambda v: (statistics.median(v), min(v), max(v)) if v else None
print(f"{key} grid {grid}: {len(sel)} launches, queues {sorted(set(r[4] for r in sel))}")
print("  duration us  med/min/max", [round(x / 1e3, 1) for x in q(dur)])
print("  period   us  med/min/max", [round(x / 1e3, 1) for x in q(per)])
print("  gap      us  med/min/max", [round(x / 1e3, 1) for x in q(gap)])
hist = collections.Counter(int(round(g / 1e3 / 5.0)) * 5 for g in gap)
print("  gap histogram (us: count)", dict(sorted(hist.items())))
# the other kernels: duration by whether they overlapped a selected launch
ov = collections.defaultdict(lambda: [[], []])
j = 0
for r in rows:
    if key in r[2]:
        continue
    inside = any(s[0] < r[1] and r[0] < s[1] for s in sel if abs(s[0] - r[0]) < 5_000_000)
    ov[r[2]][1 if inside else 0].append(r[1] - r[0])
for k, (alone, beside) in sorted(ov.items()):
    fmt = lambda v: f"{statistics.median(v) / 1e3:8.1f} us x{len(v):<5d}" if v else " " * 18
    print(f"  {k[:44]:44s} alone {fmt(alone)}  beside {fmt(beside)}")
