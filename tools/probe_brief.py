"""Where does the 75-85 ms go?  After a handle with gigabytes of device memory has been destroyed, ONE later stream
synchronisation of the next handle takes that much longer although the GPU timeline shows its kernels back to back.
This probe creates / destroys handles in the order bench.py's default run does and tries ways to absorb the stall outside
the timed region."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import ellalgo_rs_amd as pkg
from ellalgo_rs_amd import synth
torch.cuda.init()
def go(n, W, K, tag, settle=None):
    kinds, grads, b0, b1 = synth.deep_cuts(n, W + K)
    space = pkg.Ell.new_with_scalar(1.0, np.zeros(n), device=0)
    space.queue_upload(kinds, grads, b0, b1)
    space.queue_run(0, W, fused=True); space.flush()
    torch.cuda.synchronize(); space.synchronize()
    t = time.perf_counter()
    if settle == "sleep":
        time.sleep(0.3); space.synchronize()
    elif settle == "syncs":
        for _ in range(30):
            space.tsq(); space.synchronize(); time.sleep(0.01)
    elif settle == "rerun":          # a second, untimed copy of the timed work
        pass
    ts = time.perf_counter() - t
    t = time.perf_counter(); space.queue_run(W, K, fused=True); space.flush(); space.synchronize(); t3 = time.perf_counter()
    print(f"{tag:44s} settle {1e3*ts:7.1f} ms   timed {1e3*(t3-t):7.2f} ms -> {K/(t3-t):8.0f} upd/s", flush=True)
    del space
for settle in (None, "sleep", "syncs"):
    go(16384, 20, 200, f"16384 [{settle}]", settle)
    go(32768, 16, 64, f"32768 after 16384 [{settle}]", settle)
