import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import ellalgo_rs_amd as pkg
from ellalgo_rs_amd import synth
torch.cuda.init()
def go(n, W, K, tag, pre=None):
    kinds, grads, b0, b1 = synth.deep_cuts(n, W + K)
    if pre: pre()
    space = pkg.Ell.new_with_scalar(1.0, np.zeros(n), device=0)
    space.queue_upload(kinds, grads, b0, b1)
    t = time.perf_counter(); space.queue_run(0, W, fused=True); t1 = time.perf_counter(); space.flush(); t2 = time.perf_counter()
    torch.cuda.synchronize(); space.synchronize(); t3 = time.perf_counter()
    print(f"{tag} warm: run {1e3*(t1-t):.2f} flush {1e3*(t2-t1):.2f} sync {1e3*(t3-t2):.2f} ms", flush=True)
    t = time.perf_counter(); space.queue_run(W, K, fused=True); t1 = time.perf_counter(); space.flush(); t2 = time.perf_counter()
    space.synchronize(); t3 = time.perf_counter()
    print(f"{tag} timed: run {1e3*(t1-t):.2f} flush {1e3*(t2-t1):.2f} sync {1e3*(t3-t2):.2f} ms  -> {K/(t3-t):.0f} upd/s", flush=True)
    st, ts = space.queue_results()
    assert np.all(st == 0)
    del space
go(16384, 20, 200, "16384")
go(32768, 16, 64, "after 16384: 32768")
go(16384, 20, 200, "16384 again")
go(32768, 16, 64, "after 16384 + 0.5 s sleep: 32768", pre=lambda: time.sleep(0.5))
go(16384, 20, 200, "16384 again")
go(32768, 16, 64, "after 16384 + second warm-up: 32768 (W=32)") if False else None
