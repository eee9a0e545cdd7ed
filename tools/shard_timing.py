#!/usr/bin/env python3
"""Per-rank kernel times of the symmetric row-shard schedule, measured on ONE GPU: all P shards of an n x n Ell
live on the card at once, their partial GEMV vectors are summed by a torch add (standing in for the all-reduce),
and every shard's kernels are timed with the library's HIP events.  max over ranks = what one GPU of a P-GPU
node spends per update (communication excluded).    python tools/shard_timing.py 16384 8 [updates]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ellalgo_rs_amd as pkg  # noqa: E402
from ellalgo_rs_amd import synth  # noqa: E402
from ellalgo_rs_amd.sharded import HipShardEngine, partition, partition_symmetric  # noqa: E402

n, P = int(sys.argv[1]), int(sys.argv[2])
K = int(sys.argv[3]) if len(sys.argv) > 3 else 32
symmetric = os.environ.get("SHARD_SYMMETRIC", "1") != "0"
depth = int(os.environ.get("SHARD_DEPTH", "16" if symmetric else "8"))
kinds, grads, b0, b1 = synth.deep_cuts(n, K)
engines = []
for r in range(P):
    row0, nrows = (partition_symmetric if symmetric else partition)(n, P, r)
    e = HipShardEngine(n, row0, nrows, 1.0, None, None, np.zeros(n), device=0)
    if symmetric:
        e.set_symmetric(True)
    e.set_defer_depth(depth)
    engines.append(e)
for e in engines:
    e.profile_enable(True)
for i in range(K):
    for e in engines:   # one shard at a time: on a real node each has a GPU to itself
        with e.issue():
            e.begin(int(kinds[i]), grads[i], float(b0[i]), 0, 0.0)
        torch.cuda.synchronize()
    if symmetric:
        total = engines[0].gt.clone()
        for e in engines[1:]:
            total += e.gt
        for e in engines:
            e.gt.copy_(total)
    else:
        for e in engines:
            for o in engines:
                if o is not e:
                    e.gt[o.row0:o.row0 + o.nrows].copy_(o.gt[o.row0:o.row0 + o.nrows])
    torch.cuda.synchronize()
    st = []
    for e in engines:
        st.append(e.end())   # synchronous
    assert all(s == 0 for s in st), (i, st)
print(f"n={n} P={P} symmetric={symmetric} depth={depth}: per-update kernel time per rank (ms), {K} updates")
worst = 0.0
for r, e in enumerate(engines):
    pr = e.profile_read()
    per = {k: (ms / K) for k, (ms, cnt) in pr.items() if cnt}
    tot = sum(per.values())
    worst = max(worst, tot)
    print(f"  rank {r}: rows {e.nrows:6d}  total {tot:.4f}  " + "  ".join(f"{k} {v:.4f}" for k, v in per.items()))
print(f"  max over ranks {worst:.4f} ms  -> {1e3 / worst:.0f} updates/s before communication")
