"""n = 4096 resident batches: where a batch's wall time goes (queue_run incl. its synchronisation | flush | synchronize) for the
cooperative launch (ELLHIP_OPT_RESIDENT = 1, the default) and the plain launch (2), and the kernel's own time
from HIP events (profile_read)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import ellalgo_rs_amd as pkg
from ellalgo_rs_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for K in (24, 200):
    kinds, grads, b0, b1 = synth.deep_cuts(n, 20 + 4 * K)
    for mode in (1, 2):
        e = pkg.Ell.new_with_scalar(1.0, np.zeros(n))
        e.set_option(pkg.capi.OPT_RESIDENT, mode)
        e.queue_upload(kinds, grads, b0, b1)
        e.queue_run(0, 20, fused=True); e.synchronize()
        rows = []
        for r in range(4):
            if r == 3:
                e.profile_enable(True)
            t0 = time.perf_counter(); e.queue_run(20 + r * K, K, fused=True); t1 = time.perf_counter()
            e.flush(); t2 = time.perf_counter(); e.synchronize(); t3 = time.perf_counter()
            rows.append([round((b - a) * 1e3, 3) for a, b in ((t0, t1), (t1, t2), (t2, t3), (t0, t3))])
        prof = e.profile_read()
        st, _ = e.queue_results(); assert np.all(st == 0)
        print(f"K={K:4d} RESIDENT={mode}: [run, flush, sync, total] ms {rows}  kernel {prof['resident'][0]:.3f} ms -> {K / (min(x[3] for x in rows) * 1e-3):9.0f} updates/s")
