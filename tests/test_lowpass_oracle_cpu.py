"""CPU: the oracle's LowpassOracle restatement (oracle/lowpass_oracle.c) against what the reference's own
tests hold for it (src/oracles/lowpass_oracle.rs:169-240, tests/stress_tests.rs:7-24) and
against an independent numpy walk written from the reference text."""
import math

import numpy as np
import pytest

from oracle import oracle as O
from lowpass_probes import CONSTANT_SETS, negative_x0_probe, probe_points, transition_probe


def numpy_walk(spec, st, x, lp_sq, up_sq, sp_sq):
    """assess_feas (src/oracles/lowpass_oracle.rs:58-133) on precomputed row values; returns
    (kind, row, state') with kind in {'up','lp','sp','neg3','neg2','x0',None}."""
    st = dict(st)
    mdim = spec.shape[0]
    for _ in range(st["nwpass"]):
        st["idx1"] += 1
        if st["idx1"] == st["nwpass"]:
            st["idx1"] = 0
        val = seq_dot(spec[st["idx1"]], x)
        if val > up_sq:
            return "up", st["idx1"], val, st
        if val < lp_sq:
            return "lp", st["idx1"], val, st
    st["fmax"], st["kmax"] = -math.inf, -1
    for _ in range(st["nwstop"], mdim):
        st["idx3"] += 1
        if st["idx3"] == mdim:
            st["idx3"] = st["nwstop"]
        val = seq_dot(spec[st["idx3"]], x)
        if val > sp_sq:
            return "sp", st["idx3"], val, st
        if val < 0.0:
            return "neg3", st["idx3"], val, st
        if val > st["fmax"]:
            st["fmax"], st["kmax"] = val, st["idx3"]
    for _ in range(st["nwpass"], st["nwstop"]):
        st["idx2"] += 1
        if st["idx2"] == st["nwstop"]:
            st["idx2"] = st["nwpass"]
        val = seq_dot(spec[st["idx2"]], x)
        if val < 0.0:
            return "neg2", st["idx2"], val, st
    if x[0] < 0.0:
        return "x0", -1, float(x[0]), st
    return None, -1, 0.0, st


def seq_dot(a, x):
    s = 0.0
    for u, v in zip(a.tolist(), x.tolist()):  # left fold from 0.0 (src/arr.rs:443-451)
        s += u * v
    return s


def test_case_constants_as_written_are_inverted():
    wpass, wstop, lp_sq, up_sq, sp_sq = O.lowpass_case(False)
    assert (wpass, wstop) == (0.12, 0.20)
    assert lp_sq > up_sq
    assert lp_sq == pytest.approx(162.1138938, rel=1e-9) and up_sq == pytest.approx(0.006168502750, rel=1e-9)
    assert sp_sq == pytest.approx((0.125 * math.pi) ** 2, rel=1e-14)


def test_case_constants_match_the_product_mirror():
    import ellalgo_rs_amd as pkg
    for corrected in (False, True):
        assert pkg.lowpass_case_constants(corrected) == pytest.approx(O.lowpass_case(corrected), rel=1e-15)


def test_constructor_fields():
    o = O.OracleLowpass.create_case(32)
    st = o.state()
    mdim = 15 * 32
    assert st["nwpass"] == math.floor(0.12 * (mdim - 1)) + 1 == 58
    assert st["nwstop"] == math.floor(0.20 * (mdim - 1)) + 1 == 96
    assert (st["idx1"], st["idx2"], st["idx3"], st["kmax"], st["more_alt"]) == (-1, 57, 95, -1, 1)
    assert st["fmax"] == -math.inf
    sp = o.spectrum
    assert sp.shape == (mdim, 32)
    assert np.all(sp[:, 0] == 1.0) and np.all(sp[0, 1:] == 2.0)
    w = np.linspace(0.0, math.pi, mdim)
    assert np.allclose(sp[:, 1:], 2.0 * np.cos(np.outer(w, np.arange(1, 32))), rtol=0, atol=1e-13)


# ---- what the reference's own tests assert (all of it)
def test_reference_test_lowpass_oracle():  # :187-193
    o = O.OracleLowpass.create_case(32)
    assert o.assess_feas(np.zeros(32)) is not None


def test_reference_test_lowpass_oracle_direct():  # :195-205
    o = O.OracleLowpass.create_case(32)
    h = np.zeros(32)
    h[0] = 1.0
    (g, (b0, b1)), shrunk, gamma = o.assess_optim(h, o.s.sp_sq)
    assert g.size == 32 and math.isfinite(b0)


def test_reference_test_negative_transition_and_first_coeff():  # :207-239
    o = O.OracleLowpass.create_case(32)
    h = np.zeros(32)
    h[0] = -0.1
    r = o.assess_feas(h)
    assert r is not None and r[0].size == 32
    o = O.OracleLowpass.create_case(32)
    h = np.full(32, 0.01)
    h[0] = -0.5
    r = o.assess_feas(h)
    assert r is not None and np.any(r[0] != 0.0)


@pytest.mark.parametrize("n,kappa,gamma0", [(32, 40.0, None), (128, 1.0, None), (32, 1.0, 1e-12)])
def test_reference_runs_end_at_iteration_zero(n, kappa, gamma0):
    """run_lowpass (:174-185) and tests/stress_tests.rs:7-24: with the constants as written the first cut is
    ParallelCut(lp_sq, Some(up_sq)) with beta1 < beta0 -> NoSoln (src/ell_calc.rs:757-759) -> (None, 0)."""
    o = O.OracleLowpass.create_case(n)
    e = O.OracleEll.new_with_scalar(kappa, np.zeros(n))
    g0 = o.s.sp_sq if gamma0 is None else gamma0
    xb, niter, gamma, last = o.cutting_plane_optim(e, g0, 50000, 1e-14)
    assert xb is None and niter == 0 and last == O.NOSOLN and gamma == g0
    assert o.state()["idx1"] == 0  # one row visited
    o2 = O.OracleLowpass.create_case(n)
    g, (b0, b1) = o2.assess_feas(np.zeros(n))
    _, _, lp_sq, up_sq, _ = O.lowpass_case(False)
    assert (b0, b1) == (lp_sq, up_sq) and np.array_equal(g, -o2.spectrum[0])


# ---- the walk itself against an independent restatement
@pytest.mark.parametrize("n", [4, 9, 16, 24])
@pytest.mark.parametrize("cset", sorted(CONSTANT_SETS))
def test_walk_matches_numpy_restatement(n, cset):
    rng = np.random.default_rng(100 + n)
    wpass, wstop, lp_sq, up_sq, sp_sq = CONSTANT_SETS[cset]
    o = O.OracleLowpass(n, wpass, wstop, lp_sq, up_sq, sp_sq)
    spec = o.spectrum.copy()
    st = {k: v for k, v in o.state().items()}
    kinds = set()
    for it, x in enumerate(probe_points(n, rng, 240)):
        kind, row, val, st = numpy_walk(spec, st, x, lp_sq, up_sq, sp_sq)
        kinds.add(kind)
        r = o.assess_feas(x)
        got = o.state()
        for k in ("idx1", "idx2", "idx3"):
            assert got[k] == st[k], (it, k)
        if kind is None:
            assert r is None and got["more_alt"] == 0
            assert got["kmax"] == st["kmax"] and got["fmax"] == st["fmax"]
            continue
        g, (b0, b1) = r
        if kind == "up":
            assert np.array_equal(g, spec[row]) and (b0, b1) == (val - up_sq, val - lp_sq)
        elif kind == "lp":
            assert np.array_equal(g, -spec[row]) and (b0, b1) == (-val + lp_sq, -val + up_sq)
        elif kind == "sp":
            assert np.array_equal(g, spec[row]) and (b0, b1) == (val - sp_sq, val)
        elif kind == "neg3":
            assert np.array_equal(g, -spec[row]) and (b0, b1) == (-val, -val + sp_sq)
        elif kind == "neg2":
            assert np.array_equal(g, -spec[row]) and (b0, b1) == (-val, None)
        else:
            e0 = np.zeros(n)
            e0[0] = -1.0
            assert np.array_equal(g, e0) and (b0, b1) == (-x[0], None)
    if cset == "very_loose" and n >= 9:
        assert None in kinds, kinds
    if cset == "loose":
        assert len(kinds) >= 3, kinds


def test_every_return_statement_is_reached():
    n = 128
    seen = set()
    for cset, x in [("very_loose", transition_probe(n)), ("negative_passband_allowed", negative_x0_probe(n)),
                    ("very_loose", np.eye(n)[0] * 0.5), ("loose", np.eye(n)[0] * 2.0), ("loose", np.eye(n)[0] * 0.1),
                    ("loose", np.eye(n)[0]), ("very_loose", -np.eye(n)[0] * 0.0 + 1e-3 * np.cos(np.arange(n) * 2.5))]:
        c = CONSTANT_SETS[cset]
        o = O.OracleLowpass(n, *c)
        kind, row, val, st = numpy_walk(o.spectrum, o.state(), x, c[2], c[3], c[4])
        r = o.assess_feas(x)
        seen.add(kind)
        assert (r is None) == (kind is None)
        if kind in ("neg2", "x0"):
            assert r[1][1] is None
        assert o.state()["idx1"] == st["idx1"] and o.state()["idx2"] == st["idx2"] and o.state()["idx3"] == st["idx3"]
    assert {"neg2", "x0", "up", "lp", "sp", None} <= seen, seen


def test_corrected_case_solves_and_shrinks_gamma():
    """parity-unpinned case (no reference answer): sanity of the restatement itself."""
    n = 32
    o = O.OracleLowpass.create_case(n, corrected=True)
    e = O.OracleEll.new_with_scalar(40.0, np.zeros(n))
    g0 = o.s.sp_sq
    xb, niter, gamma, last = o.cutting_plane_optim(e, g0, 50000, 1e-14)
    assert xb is not None and 100 < niter < 50000 and gamma < g0
    # x_best satisfies the constraints at the returned gamma (up to the walk's own tolerance)
    vals = o.spectrum @ xb
    st = o.state()
    _, _, lp_sq, up_sq, _ = O.lowpass_case(True)
    assert np.all(vals[:st["nwpass"]] <= up_sq + 1e-9) and np.all(vals[:st["nwpass"]] >= lp_sq - 1e-9)
    assert np.all(vals[st["nwstop"]:] <= gamma + 1e-9) and np.all(vals >= -1e-9)
