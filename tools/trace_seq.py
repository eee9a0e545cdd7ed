#!/usr/bin/env python3
"""Print the kernel sequence (start offset, duration, gap to the previous kernel, name, grid) of a rocprofv3 kernel trace
between two launches of a marker kernel.  usage: trace_seq.py <dir> <marker substring> <grid_x workgroups> <from-th> <to-th>"""
import csv
import glob
import os
import sys

d, marker, gx, a, b = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
rows = []
for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        short = name.split("ellhip::")[1].split("(")[0] if "ellhip::" in name else name[:40]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]),
                     int(r["Grid_Size_Y"])))
rows.sort()
idx = [i for i, r in enumerate(rows) if marker in r[2] and r[3] == gx]
lo, hi = idx[a], idx[min(b, len(idx) - 1)]
t0 = rows[lo][0]
prev_end = rows[lo][0]
for r in rows[lo:hi + 1]:
    print(f"{(r[0] - t0) / 1e3:10.1f} us  dur {(r[1] - r[0]) / 1e3:9.1f}  gap {(r[0] - prev_end) / 1e3:9.1f}  {r[2][:50]} ({r[3]},{r[4]})")
    prev_end = r[1]
