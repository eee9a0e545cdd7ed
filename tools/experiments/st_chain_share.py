"""How much of an EllStable persistent solve is the diagonal-block chain arithmetic?

Run twice on a GPU box: with the product build, and with a measurement build whose chains take fewer steps
(ELLHIP_EXTRA_HIPCC_FLAGS=-DST_EXP_CHAIN_STEPS=k python -m ellalgo-rs_amd.build; results are then wrong, timing is not).
Prints the per-launch time of the forward / backward solves from the handle's own HIP events.
"""
import importlib
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("ellalgo-rs_amd")


def main():
    for n in (4096, 16384):
        steps = 60
        kinds, grads, b0, b1 = pkg.synth.deep_cuts(n, steps)
        space = pkg.EllStable.new_with_matrix(1.0, pkg.synth.stable_factor(n), np.zeros(n))
        space.queue_upload(kinds, grads, b0, b1)
        space.queue_run(0, 10)
        space.synchronize()
        import time
        t0 = time.perf_counter()
        space.queue_run(10, 25)
        space.synchronize()
        wall = (time.perf_counter() - t0) / 25
        space.profile_enable(True)
        space.queue_run(35, steps - 35)
        space.synchronize()
        prof = space.profile_read()
        status, _ = space.queue_results()
        nb = (n + 127) // 128
        out = {k: v for k, v in prof.items() if v[1] > 0}
        line = ", ".join(f"{k} {v[0] / v[1] * 1e3:.1f} us ({v[0] / v[1] * 1e3 / nb:.2f} us/block)" if k.startswith("stable_") and k != "stable_factor"
                         else f"{k} {v[0] / v[1] * 1e3:.1f} us" for k, v in out.items())
        print(f"n={n}: {1.0 / wall:.0f} updates/s ({wall * 1e6:.0f} us per update); {line}; statuses ok={int((status[:steps] == 0).sum())}/{steps}", flush=True)


if __name__ == "__main__":
    main()
