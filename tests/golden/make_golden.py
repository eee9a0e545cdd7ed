#!/usr/bin/env python3
"""Generates tests/golden/ell_dense_n*.npz.  Run in the BUILD container only (needs /root/reference).

What is cross-checked: the dense arithmetic of Ell::update_core (GEMV, omega, xc axpy, rank-1, kappa,
no_defer_trick) as implemented by the reference's in-tree Python sibling,
`python_ai/ellalgo/ell.py::Ell._update_core` (numpy: `mq @ grad`, outer product).  That sibling's
EllCalc and constructors are NOT faithful to the Rust crate, so only `_update_core` is
used, driven with a `cut_strategy` that returns the (status, rho, sigma, delta) of the oracle's
restatement of the Rust EllCalc (itself pinned by the Rust known answers in tests/test_oracle_pins.py).

The files hold plain data: the cut sequence (inputs) and the reference's state after every step
(outputs).  PYTHONDONTWRITEBYTECODE keeps the read-only reference tree untouched.
"""
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference/python_ai")

import numpy as np  # noqa: E402
from ellalgo.cutting_plane import CutStatus as RefStatus  # noqa: E402  (reference, python_ai)
from ellalgo.ell import Ell as RefEll  # noqa: E402

from oracle import oracle  # noqa: E402
from util import mixed_cut  # noqa: E402

REF_STATUS = {0: RefStatus.SUCCESS, 1: RefStatus.NO_SOLN, 2: RefStatus.NO_EFFECT, 3: RefStatus.UNKNOWN}


def q_cut(i, tau, n, rng):
    """Discrete-q sequence (update_q only) that reaches every return of calc_parallel_q / calc_bias_cut_q
    (src/ell_calc.rs:787-812, 892-908): NoEffect through eta = tsq + n b0 b1 <= 0 and through eta = tau + n beta < 0,
    NoSoln (beta1 < beta0; tau < beta), the fall-back to the single cut (beta1^2 >= tsq) and both Success forms."""
    sel = i % 8
    if sel == 0:
        return 2, -0.01 * tau * rng.random(), tau * (0.2 + 0.5 * rng.random())   # parallel, eta > 0: Success
    if sel == 1:
        return 2, -tau * (0.3 + 0.3 * rng.random()), tau * (0.5 + 0.3 * rng.random())  # parallel, eta <= 0: NoEffect
    if sel == 2:
        return 2, 0.1 * tau * rng.random(), None                                  # single, Success
    if sel == 3:
        return 2, -tau * (2.0 / n + 0.2 * rng.random()), None                     # single, eta < 0: NoEffect
    if sel == 4:
        return 2, 0.05 * tau, tau * (1.0 + rng.random())                          # beta1^2 >= tsq: single cut with beta0
    if sel == 5:
        return 2, 0.3 * tau, 0.1 * tau                                            # beta1 < beta0: NoSoln
    if sel == 6:
        return 2, -tau * (0.5 + 0.2 * rng.random()), tau * (1.2 + rng.random())   # fall-back, then eta < 0: NoEffect
    return 2, 1.5 * tau, None                                                     # tau < beta: NoSoln


def generate(n: int, k: int, seed: int, no_defer: bool, cuts=None):
    rng = np.random.default_rng(seed)
    kappa0 = 2.0
    xc0 = np.linspace(-1.0, 1.0, n)
    ref = RefEll.new_with_matrix(kappa0, np.eye(n), xc0.copy())
    ref.no_defer_trick = no_defer
    calc = oracle.Calc(n)
    kinds, grads, b0s, b1s, statuses = [], [], [], [], []
    xcs, kappas, tsqs = [], [], []
    for i in range(k):
        g = rng.standard_normal(n)
        g /= np.linalg.norm(g)
        tau = float(np.sqrt(max(ref.kappa * (g @ (ref.mq @ g)), 0.0)))
        kind, b0, b1 = mixed_cut(i, g, tau, rng) if cuts is None else cuts(i, tau, n, rng)
        st = ref._update_core(g, (b0, b1), lambda beta, tsq: (
            lambda r: (REF_STATUS[r[0]], r[1]))(calc.dispatch(kind, beta[0], beta[1], tsq)))
        kinds.append(kind), grads.append(g), b0s.append(b0), b1s.append(np.nan if b1 is None else b1)
        statuses.append({v: k_ for k_, v in REF_STATUS.items()}[st])
        xcs.append(ref.xc.copy()), kappas.append(ref.kappa), tsqs.append(ref.tsq)
    return dict(n=n, kappa0=kappa0, xc0=xc0, no_defer=int(no_defer), kinds=np.array(kinds, dtype=np.int32),
                grads=np.array(grads), beta0=np.array(b0s), beta1=np.array(b1s),
                status=np.array(statuses, dtype=np.int32), xc=np.array(xcs), kappa=np.array(kappas),
                tsq=np.array(tsqs), mq_final=ref.mq.copy())


if __name__ == "__main__":
    for n, k, nd in [(4, 24, False), (16, 40, False), (64, 48, False), (16, 24, True)]:
        d = generate(n, k, seed=1000 + n + int(nd), no_defer=nd)
        name = os.path.join(HERE, f"ell_dense_n{n}{'_nodefer' if nd else ''}.npz")
        np.savez_compressed(name, **d)
        print(name, "statuses:", np.bincount(d["status"], minlength=3))
    for n, k in [(16, 48)]:
        d = generate(n, k, seed=2000 + n, no_defer=False, cuts=q_cut)
        name = os.path.join(HERE, f"ell_dense_n{n}_qcuts.npz")
        np.savez_compressed(name, **d)
        print(name, "statuses:", np.bincount(d["status"], minlength=3))
        assert np.all(np.bincount(d["status"], minlength=3) > 0)   # Success, NoSoln AND NoEffect all occur
