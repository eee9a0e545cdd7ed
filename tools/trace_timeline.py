#!/usr/bin/env python3
"""From a rocprofv3 kernel trace (+ memory-copy trace when present) of a LIVE loop: the median timeline of one iteration --
every event between two consecutive launches of the anchor kernel (default k_symv), with its offset from the anchor's start,
its duration and the idle gap in front of it.

usage: trace_timeline.py <dir with *_kernel_trace.csv [*_memory_copy_trace.csv]> [anchor substring]"""
import collections
import csv
import glob
import os
import statistics
import sys

d = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "k_symv<"
ev = []
for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        short = name.split("ellhip::")[1].split("(")[0] if "ellhip::" in name else name[:40]
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short))
for f in glob.glob(os.path.join(d, "**", "*_memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy:" + r.get("Direction", "?")))
ev.sort()
anchors = [i for i, e in enumerate(ev) if key in e[2]]
# steady state: the middle half of the anchors
lo, hi = len(anchors) // 4, 3 * len(anchors) // 4
per = []
slots = collections.defaultdict(lambda: {"off": [], "dur": [], "gap": []})
shape = collections.Counter()
for a, b in zip(anchors[lo:hi], anchors[lo + 1:hi + 1]):
    seq = ev[a:b]
    shape[tuple(e[2] for e in seq)] += 1
common = shape.most_common(1)[0][0]
for a, b in zip(anchors[lo:hi], anchors[lo + 1:hi + 1]):
    seq = ev[a:b]
    if tuple(e[2] for e in seq) != common:
        continue
    per.append(ev[b][0] - ev[a][0])
    prev_end = None
    for j, e in enumerate(seq):
        s = slots[j]
        s["off"].append(e[0] - seq[0][0])
        s["dur"].append(e[1] - e[0])
        s["gap"].append(0 if prev_end is None else e[0] - prev_end)
        prev_end = e[1]
    slots[len(seq)]["off"].append(ev[b][0] - seq[0][0])
    slots[len(seq)]["dur"].append(0)
    slots[len(seq)]["gap"].append(ev[b][0] - prev_end)
med = lambda v: statistics.median(v) / 1e3
print(f"anchor {key}: {len(anchors)} launches; iteration shapes: {len(shape)}; the common one covers {shape[common]} of {hi - lo}")
print(f"period us: median {med(per):.1f}  min {min(per) / 1e3:.1f}  max {max(per) / 1e3:.1f}")
print(f"{'event':46s} {'start +us':>10s} {'gap before':>11s} {'duration':>9s}")
for j in range(len(common) + 1):
    name = common[j] if j < len(common) else "(next " + key + ")"
    s = slots[j]
    print(f"{name[:46]:46s} {med(s['off']):10.1f} {med(s['gap']):11.1f} {med(s['dur']):9.1f}")
busy = sum(med(slots[j]["dur"]) for j in range(len(common)))
print(f"GPU busy {busy:.1f} us of {med(per):.1f} us per iteration; idle {med(per) - busy:.1f} us")
