#!/usr/bin/env python3
"""Which deferred depth pays at which size (one GPU): synchronous ellhip_update calls/s and pipelined queue
updates/s for depth 1 and 8 (and 16 where the lower-triangle schedule exists)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ellalgo_rs_amd as pkg  # noqa: E402
from ellalgo_rs_amd import synth  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048, 3072, 4096, 6144]:
    k = 400
    kinds, grads, b0, b1 = synth.deep_cuts(n, 2 * k)
    row = [f"n={n:6d}"]
    lower = n % 2 == 0 and n >= pkg.capi.default_option(pkg.capi.OPT_SYMV_MIN_N)
    for depth in ((1, 8, 16) if lower else (1, 8)):
        e = pkg.Ell.new_with_scalar(1.0, np.zeros(n))
        e.defer_depth = depth
        for i in range(40):
            e.update_bias_cut((grads[i], float(b0[i])))
        e.flush(); e.synchronize()
        t = time.perf_counter()
        for i in range(40, k):
            e.update_bias_cut((grads[i], float(b0[i])))
        e.flush(); e.synchronize()
        host = (k - 40) / (time.perf_counter() - t)
        e.queue_upload(kinds[k:], grads[k:], b0[k:], b1[k:])
        e.queue_run(0, 40, fused=True); e.flush(); e.synchronize()
        t = time.perf_counter()
        e.queue_run(40, k - 40, fused=True); e.flush(); e.synchronize()
        q = (k - 40) / (time.perf_counter() - t)
        st, _ = e.queue_results()
        assert np.all(st == 0)
        row.append(f"d{depth}: host {host:8.0f}/s queue {q:8.0f}/s")
    print("   ".join(row), flush=True)
