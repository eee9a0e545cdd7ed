import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import ellalgo_rs_amd as gpu
from util import random_factor
capi = gpu.capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 129
f = random_factor(n, 271 + n)
def mk(solve, factor):
    capi.set_default_option(capi.OPT_STABLE_SOLVE, solve); capi.set_default_option(capi.OPT_STABLE_FACTOR, factor)
    return gpu.EllStable.new_with_matrix(1.0, f, np.zeros(n))
hs = {"b00": mk(0, 0), "c11": mk(1, 1), "e22": mk(2, 2), "m3": mk(3, 2), "p10": mk(1, 0), "q01": mk(0, 1)}
rng = np.random.default_rng(13 * n)
for i in range(14):
    gr = rng.standard_normal(n); gr /= np.linalg.norm(gr)
    beta = 5.0 if i in (5, 10) else 0.05 * rng.random()
    st = {k: int(h.update_bias_cut((gr, beta))) for k, h in hs.items()}
    ts = {k: h.tsq() for k, h in hs.items()}
    ref = ts["b00"]
    print(i, st, {k: (0 if v == ref else (v - ref) / ref) for k, v in ts.items()})
