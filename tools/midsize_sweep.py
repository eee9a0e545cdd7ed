#!/usr/bin/env python3
"""Sizes between the resident kernel's range (n <= 4224) and the default threshold of the lower-triangle schedule (8192):
synchronous ellhip_update calls/s and pipelined queue updates/s with the full-row schedule at depth 8 (today's default
there) and with the lower-triangle schedule at depth 24 (ELLHIP_OPT_SYMV_MIN_N lowered), to place the threshold."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ellalgo_rs_amd as pkg  # noqa: E402
from ellalgo_rs_amd import synth  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [4288, 5120, 6144, 7168, 8128]:
    k = 400
    kinds, grads, b0, b1 = synth.deep_cuts(n, 2 * k)
    row = [f"n={n:6d}"]
    for name, min_n, depth in (("full-row d8", 65536, 8), ("lower-triangle d24", 512, 24)):
        pkg.capi.set_default_option(pkg.capi.OPT_SYMV_MIN_N, min_n)
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
        time.sleep(0.3); e.synchronize()
        t = time.perf_counter()
        e.queue_run(40, k - 40, fused=True); e.flush(); e.synchronize()
        q = (k - 40) / (time.perf_counter() - t)
        st, _ = e.queue_results()
        assert np.all(st == 0)
        row.append(f"{name}: calls {host:7.0f}/s queue {q:8.0f}/s")
        del e
    print("   ".join(row), flush=True)
pkg.capi.set_default_option(pkg.capi.OPT_SYMV_MIN_N, 5120)
