import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import ellalgo_rs_amd as gpu
from util import random_factor
capi = gpu.capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 129
variant = sys.argv[2] if len(sys.argv) > 2 else "full"
f = random_factor(n, 271 + n)
def mk(solve, factor):
    capi.set_default_option(capi.OPT_STABLE_SOLVE, solve); capi.set_default_option(capi.OPT_STABLE_FACTOR, factor)
    return gpu.EllStable.new_with_matrix(1.0, f, np.zeros(n))
a = mk(3, 2); b = mk(0, 0); c = mk(1, 1); d = mk(2, 2); e2 = mk(2, 2); e3 = mk(3, 2); d3 = mk(3, 2)
forms = [(0, 0), (1, 0), (1, 1), (2, 1), (2, 2), (0, 1), (2, 0), (1, 2), (2, 2), (0, 0), (1, 1), (2, 0), (1, 2), (2, 1)]
forms3 = [(0, 0), (1, 0), (3, 1), (3, 1), (3, 2), (3, 0), (2, 1), (2, 2), (3, 2), (0, 1), (3, 0), (3, 0), (1, 2), (3, 1)]
rng = np.random.default_rng(13 * n)
for i in range(14):
    gr = rng.standard_normal(n); gr /= np.linalg.norm(gr)
    beta = 5.0 if i in (5, 10) else 0.05 * rng.random()
    if variant != "nowalk":
        for h, fm in ((d, forms[i]), (d3, forms3[i])):
            h.set_option(capi.OPT_STABLE_SOLVE, fm[0]); h.set_option(capi.OPT_STABLE_FACTOR, fm[1])
    hs = dict(a=a, b=b, c=c, d=d, e2=e2, e3=e3, d3=d3)
    st = {k: int(h.update_bias_cut((gr, beta))) for k, h in hs.items()}
    ts = {k: h.tsq() for k, h in hs.items()}
    ref = ts["b"]
    print(i, list(st.values()), {k: (0 if v == ref else float("%.2e" % ((v - ref) / ref))) for k, v in ts.items()})
    if variant != "noobs" and i in (3, 5, 8):
        m = e3.mq; mb = b.mq
        print("   obs", np.max(np.abs(np.triu(m,1)-np.triu(mb,1))), np.max(np.abs(np.tril(m,-1)-np.tril(mb,-1))), np.max(np.abs(np.diag(m)-np.diag(mb))))
