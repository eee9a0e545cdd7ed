"""Shared helpers for the parity tests."""
import numpy as np

TOL = 1e-10  # BASELINE.json north_star: (xc, Q, kappa) within 1e-10 relative


def rel_inf(a, b):
    """max|a-b| / max|b| (inf-norm relative); exact 0 when both are all-zero."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.max(np.abs(a - b)) if a.size else 0.0
    s = np.max(np.abs(b)) if b.size else 0.0
    if d == 0.0:
        return 0.0
    return d / s if s > 0 else np.inf


def assert_state_close(gpu_space, orc_space, tol=TOL, what=""):
    """SURVEY 8d parity check: Q, xc (relative inf-norm, abs fallback when xc ~ 0), kappa, tsq."""
    q_err = rel_inf(gpu_space.mq, orc_space.mq)
    assert q_err <= tol, f"{what} Q rel err {q_err}"
    xg, xo = gpu_space.xc(), np.array(orc_space.xc)
    x_abs = np.max(np.abs(xg - xo)) if xo.size else 0.0
    assert x_abs <= tol * np.max(np.abs(xo)) + 1e-300, f"{what} xc abs err {x_abs}"
    ko = orc_space.kappa
    assert abs(gpu_space.kappa - ko) <= tol * abs(ko), f"{what} kappa {gpu_space.kappa} vs {ko}"
    to = orc_space.tsq
    assert abs(gpu_space.tsq() - to) <= tol * abs(to) + 1e-300, f"{what} tsq {gpu_space.tsq()} vs {to}"


def mixed_cut(i, g, tau, rng):
    """Cut number i of a sequence that exercises every EllCalc entry point; beta is scaled by the
    current tau = sqrt(kappa * g'Qg) so the sequence stays well-posed while the ellipsoid shrinks.
    Returns (kind, b0, b1_or_None)."""
    sel = i % 8
    if sel == 0:
        return 0, 0.3 * tau * rng.random(), None                       # bias, SingleCut
    if sel == 1:
        return 1, 0.0, None                                            # central, SingleCut
    if sel == 2:
        b0 = 0.1 * tau * rng.random()
        return 0, b0, b0 + tau * (0.1 + 0.5 * rng.random())            # bias, ParallelCut
    if sel == 3:
        return 1, 0.0, tau * (0.1 + 0.6 * rng.random())                # central, ParallelCut
    if sel == 4:
        return 2, 0.2 * tau * rng.random(), None                       # q, SingleCut
    if sel == 5:
        b0 = 0.1 * tau * rng.random()
        return 2, b0, b0 + tau * (0.1 + 0.5 * rng.random())            # q, ParallelCut
    if sel == 6:
        return 0, -0.1 * tau * rng.random(), tau * (1.0 + rng.random())  # parallel falls back to deep cut
    return 0, 1.5 * tau, None                                          # NoSoln: state must stay intact


def oracle_update(orc_space, kind, g, b0, b1):
    """The oracle's update: for Ell from n = 1024 up its row-parallel loop (`update_rowwise_mt`: bit-identical to the reference's
    loop order -- tests/test_oracle_pins.py compares the three forms bit for bit at n = 37, 257 and 2048 -- without the
    column-strided mirror stores that make the reference order take 0.7 s per update at n = 8192)."""
    if type(orc_space).__name__ == "OracleEll" and orc_space.n >= 1024:
        return orc_space.update_rowwise_mt(kind, g, b0, b1)
    return orc_space.update(kind, g, b0, b1)


def run_mixed(gpu_space, orc_space, k, seed, g_scale=1.0, check_every=0, tol=TOL):
    """Drive both engines with the same adaptive cut sequence; statuses must agree at every step."""
    n = orc_space.n
    rng = np.random.default_rng(seed)
    nsucc = 0
    for i in range(k):
        g = rng.standard_normal(n)
        g *= g_scale / np.linalg.norm(g)
        tau = float(np.sqrt(max(orc_space.kappa * (g @ (orc_space.mq @ g)), 0.0)))
        kind, b0, b1 = mixed_cut(i, g, tau, rng)
        so = oracle_update(orc_space, kind, g, b0, b1)
        sg = gpu_space._update(kind, (g, beta_of(b0, b1)))
        assert int(sg) == int(so), f"step {i}: status gpu={int(sg)} oracle={so}"
        nsucc += int(so == 0)
        if check_every and (i + 1) % check_every == 0:
            assert_state_close(gpu_space, orc_space, tol, what=f"step {i}")
    return nsucc


def beta_of(b0, b1):
    return float(b0) if b1 is None else (float(b0), float(b1))


def dense_spd(n, seed, rank=32, amp=0.3):
    """A dense symmetric positive definite start matrix, symmetric to the bit: I + A A' with A = amp * randn(n, rank).
    Every entry is non-zero, g'Qg stays O(1) for unit g (1 + |A'g|^2 ~ 1 + amp^2 rank), so cut sequences scaled for
    Q0 = I (tau ~ 1) stay well-posed.  (Round 3's fast paths were only ever tested from Q0 = I, where the first group of
    a queue run multiplies the identity and the off-diagonal tiles see data only after the first apply pass.)"""
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((n, rank)) * amp
    q = a @ a.T
    q[np.arange(n), np.arange(n)] += 1.0
    return np.ascontiguousarray(0.5 * (q + q.T))


def random_factor(n, seed, junk=True):
    """Packed EllStable state with a NON-trivial factor (src/ell_stable.rs:18-27 accepts any matrix): positive random
    diagonal, unit-upper-triangular factor with entries ~ 0.1/sqrt(n), and junk in the scratch triangle (the forward
    solve overwrites every scratch entry before anything reads it, src/ell_stable.rs:61-69)."""
    rng = np.random.default_rng(seed)
    m = rng.standard_normal((n, n)) * (0.1 / np.sqrt(n))
    if not junk:
        m = np.triu(m)
    m[np.arange(n), np.arange(n)] = 0.5 + rng.random(n)
    return np.ascontiguousarray(m)


def stable_tau(orc_space, g):
    """tau = sqrt(tsq) the NEXT cut with gradient g will see: from a clone of the oracle (its packed buffer is not the
    shape matrix, so g'Mg means nothing for EllStable)."""
    c = orc_space.clone()
    c.update(1, g, 0.0, None)   # a central cut always succeeds and sets tsq = kappa * omega
    return float(np.sqrt(max(c.tsq, 0.0)))


def run_mixed_stable(gpu_space, orc_space, k, seed, check_every=0, tol=TOL):
    """run_mixed for EllStable with a non-trivial factor: the same adaptive sequence over all six EllCalc entry
    points (incl. a NoSoln cut every 8th step), tau taken from the oracle itself."""
    n = orc_space.n
    rng = np.random.default_rng(seed)
    nsucc = 0
    for i in range(k):
        g = rng.standard_normal(n)
        g /= np.linalg.norm(g)
        tau = stable_tau(orc_space, g)
        kind, b0, b1 = mixed_cut(i, g, tau, rng)
        so = orc_space.update(kind, g, b0, b1)
        sg = gpu_space._update(kind, (g, beta_of(b0, b1)))
        assert int(sg) == int(so), f"step {i}: status gpu={int(sg)} oracle={so}"
        assert abs(gpu_space.tsq() - orc_space.tsq) <= tol * abs(orc_space.tsq) + 1e-300, f"step {i}: tsq"
        nsucc += int(so == 0)
        if check_every and (i + 1) % check_every == 0:
            assert_state_close(gpu_space, orc_space, tol, what=f"step {i}")
    return nsucc


def set_default(name: str, value: int) -> None:
    """ellhip_set_default_option(ELLHIP_OPT_<name>, value): what handles created from now on start with (conftest.py puts
    the factory defaults back after every test)."""
    import ellalgo_rs_amd as pkg
    pkg.capi.set_default_option(getattr(pkg.capi, "OPT_" + name), int(value))
