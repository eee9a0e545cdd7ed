"""GPU: the device-side LowpassOracle (include/ellhip_lowpass.h) and the device-resident cutting-plane loops
against the CPU oracle (oracle/lowpass_oracle.c), through the C ABI.

* assess_feas / assess_optim: same cut at the same row (gradient bit for bit, beta to rounding), same cursor
  state, for probe points that reach every return statement of src/oracles/lowpass_oracle.rs:58-150;
* device-resident cutting_plane_optim / cutting_plane_feas: the reference runs (constants as written: NoSoln at
  iteration 0) and the corrected case against the oracle loop."""
import math

import numpy as np
import pytest

from lowpass_probes import CONSTANT_SETS, negative_x0_probe, probe_points, transition_probe
from util import assert_state_close, rel_inf, set_default

pytestmark = pytest.mark.gpu

BETA_TOL = 1e-12  # row.x by a butterfly vs a left fold: a few ulps of |val| <= ~30


def beta_close(a, b, scale):
    if a is None or b is None:
        return a is None and b is None
    return abs(a - b) <= BETA_TOL * max(1.0, scale)


def compare_call(dev_o, cpu_o, x, what):
    rd = dev_o.assess_feas(x)
    rc = cpu_o.assess_feas(x)
    assert (rd is None) == (rc is None), what
    sd, sc = dev_o.state(), cpu_o.state()
    for k in ("more_alt", "idx1", "idx2", "idx3", "kmax"):
        assert sd[k] == sc[k], (what, k, sd, sc)
    assert sd["fmax"] == sc["fmax"] or abs(sd["fmax"] - sc["fmax"]) <= BETA_TOL * max(1.0, abs(sc["fmax"])), what
    if rc is None:
        return None
    gd, cut = rd
    gc, (b0, b1) = rc
    assert np.array_equal(gd, gc), what
    scale = float(np.sum(np.abs(x))) * 2.0
    assert beta_close(cut.beta0, b0, scale) and beta_close(cut.beta1, b1, scale), (what, cut, b0, b1)
    return rc


@pytest.mark.parametrize("n", [4, 9, 32, 33, 64, 128, 256])
@pytest.mark.parametrize("cset", sorted(CONSTANT_SETS))
def test_assess_feas_matches_oracle(gpu, orc, n, cset):
    c = CONSTANT_SETS[cset]
    dev_o = gpu.LowpassOracle(n, *c)
    cpu_o = orc.OracleLowpass(n, *c)
    assert np.array_equal(dev_o.spectrum, cpu_o.spectrum)
    sd, sc = dev_o.state(), cpu_o.state()
    assert sd == pytest.approx(sc)
    rng = np.random.default_rng(7 + n)
    nnone = nsingle = 0
    for it, x in enumerate(probe_points(n, rng, 100)):
        r = compare_call(dev_o, cpu_o, x, f"{cset} n={n} call {it}")
        nnone += r is None
        nsingle += r is not None and r[1][1] is None
    if cset == "very_loose" and n >= 64:
        assert nnone > 0 and nsingle > 0


def test_every_return_statement_on_the_device(gpu, orc):
    n = 128
    for cset, x in [("very_loose", transition_probe(n)), ("negative_passband_allowed", negative_x0_probe(n)),
                    ("very_loose", np.eye(n)[0] * 0.5), ("loose", np.eye(n)[0] * 2.0), ("loose", np.eye(n)[0] * 0.1),
                    ("loose", np.eye(n)[0])]:
        c = CONSTANT_SETS[cset]
        compare_call(gpu.LowpassOracle(n, *c), orc.OracleLowpass(n, *c), x, cset)


@pytest.mark.parametrize("grid", [1, 3, 7])
def test_early_exit_across_grid_rounds(gpu, orc, grid, monkeypatch):
    """A tiny grid forces many rounds of 16-row chunks per workgroup: the first violation must still be the one
    the sequential walk finds, wherever it sits."""
    set_default("LP_GRID", grid)
    n = 64
    c = CONSTANT_SETS["very_loose"]
    dev_o = gpu.LowpassOracle(n, *c)
    cpu_o = orc.OracleLowpass(n, *c)
    rng = np.random.default_rng(grid)
    for it, x in enumerate(probe_points(n, rng, 60)):
        compare_call(dev_o, cpu_o, x, f"grid={grid} call {it}")


@pytest.mark.parametrize("n,grid", [(9, 2), (33, 5), (64, 1), (64, 16), (128, 1024)])
def test_wide_scan_kernel_at_small_sizes(gpu, orc, n, grid, monkeypatch):
    """k_lp_scan_wide (the n >= 1024 kernel: 4 positions per workgroup step, columns split over the waves)
    forced on where the oracle is cheap, over several rounds of chunks."""
    set_default("LP_WIDE", 1)
    set_default("LP_GRID", grid)
    rng = np.random.default_rng(grid + n)
    for cset in ("very_loose", "loose", "corrected"):
        c = CONSTANT_SETS[cset]
        dev_o = gpu.LowpassOracle(n, *c)
        cpu_o = orc.OracleLowpass(n, *c)
        for it, x in enumerate(probe_points(n, rng, 40)):
            compare_call(dev_o, cpu_o, x, f"wide {cset} n={n} grid={grid} call {it}")


@pytest.mark.parametrize("n", [1024, 2048])
def test_assess_optim_large(gpu, orc, n):
    c = CONSTANT_SETS["corrected"]
    dev_o = gpu.LowpassOracle(n, *c)
    cpu_o = orc.OracleLowpass(n, *c)
    rng = np.random.default_rng(n)
    gamma_d = gamma_c = c[4]
    for it, x in enumerate(probe_points(n, rng, 12)):
        (gd, cd), shd, gamma_d = dev_o.assess_optim(x, gamma_d)
        (gc, (b0, b1)), shc, gamma_c = cpu_o.assess_optim(x, gamma_c)
        assert shd == shc and np.array_equal(gd, gc), it
        assert beta_close(cd.beta0, b0, 30.0) and beta_close(cd.beta1, b1, 30.0)
        assert abs(gamma_d - gamma_c) <= BETA_TOL * max(1.0, abs(gamma_c))
        sd, sc = dev_o.state(), cpu_o.state()
        assert (sd["idx1"], sd["idx2"], sd["idx3"]) == (sc["idx1"], sc["idx2"], sc["idx3"])


def test_caller_supplied_spectrum(gpu, orc):
    n = 16
    c = CONSTANT_SETS["loose"]
    cpu_o = orc.OracleLowpass(n, *c)
    spec = cpu_o.spectrum.copy()
    spec[5] *= 1.5  # not the computed table any more
    dev_o = gpu.LowpassOracle(n, *c, spectrum=spec)
    assert np.array_equal(dev_o.spectrum, spec)
    with pytest.raises(gpu.capi.EllHipError):
        gpu.LowpassOracle(n, 0.3, 0.2, 0.5, 1.5, 0.3)  # wpass > wstop
    with pytest.raises(ValueError):
        dev_o.assess_feas(np.zeros(n + 1))


# ---- device-resident loops ---------------------------------------------------------------------------

def make_spaces(gpu, orc, variant, n, kappa, depth):
    x0 = np.zeros(n)
    if variant == "ell":
        g = gpu.Ell.new_with_scalar(kappa, x0)
        g.defer_depth = depth
        return g, orc.OracleEll.new_with_scalar(kappa, x0)
    return gpu.EllStable.new_with_scalar(kappa, x0), orc.OracleEllStable.new_with_scalar(kappa, x0)


@pytest.mark.parametrize("variant,depth", [("ell", 1), ("ell", 8), ("stable", 1)])
@pytest.mark.parametrize("n,kappa,gamma0", [(32, 40.0, None), (128, 1.0, None), (32, 1.0, 1e-12)])
def test_reference_runs_end_at_iteration_zero(gpu, orc, variant, depth, n, kappa, gamma0):
    """run_lowpass (src/oracles/lowpass_oracle.rs:174-185), tests/stress_tests.rs:7-24: (None, 0), NoSoln."""
    dev_o = gpu.create_lowpass_case(n)
    cpu_o = orc.OracleLowpass.create_case(n)
    g, o = make_spaces(gpu, orc, variant, n, kappa, depth)
    g0 = cpu_o.s.sp_sq if gamma0 is None else gamma0
    xb, niter, gamma = dev_o.cutting_plane_optim(g, g0, 50000, 1e-14)
    xbo, nitero, gammao, last = cpu_o.cutting_plane_optim(o, g0, 50000, 1e-14)
    assert xb is None and xbo is None and niter == nitero == 0 and gamma == gammao == g0 and last == orc.NOSOLN
    assert dev_o.state()["idx1"] == cpu_o.state()["idx1"] == 0
    assert_state_close(g, o, what="after NoSoln at iteration 0")
    # the space is usable afterwards
    grad = np.ones(n)
    assert int(g.update_central_cut((grad, 0.0))) == o.update_central_cut(grad) == 0
    assert_state_close(g, o, what="update after the loop")


def cpu_lockstep(orc, variant, n, c, kappa, iters, perturb):
    """The oracle loop, optionally started from a Q with ONE entry moved by one ulp: how far such a run drifts
    from the unperturbed one is the yardstick for how far any differently-rounded implementation may drift."""
    o = orc.OracleLowpass(n, *c)
    x0 = np.zeros(n)
    e = (orc.OracleEll if variant == "ell" else orc.OracleEllStable).new_with_scalar(kappa, x0)
    if perturb:
        e.mq[1, 1] = np.nextafter(e.mq[1, 1], np.inf)
    xb, niter, gamma, last = o.cutting_plane_optim(e, c[4], iters, 1e-14)
    return o, e, xb, niter, gamma, last


def sensitivity(orc, variant, n, c, kappa, iters):
    _, e0, _, n0, g0, _ = cpu_lockstep(orc, variant, n, c, kappa, iters, False)
    _, e1, _, n1, g1, _ = cpu_lockstep(orc, variant, n, c, kappa, iters, True)
    if n0 != n1:
        return np.inf
    return max(rel_inf(e1.mq, e0.mq), rel_inf(np.array(e1.xc), np.array(e0.xc)), abs(e1.tsq - e0.tsq) / abs(e0.tsq),
               abs(g1 - g0) / abs(g0))


@pytest.mark.parametrize("variant,depth", [("ell", 1), ("ell", 8), ("stable", 1)])
@pytest.mark.parametrize("n,max_iters", [(16, 10), (32, 16), (32, 64), (32, 65), (48, 150), (64, 300)])
def test_optim_loop_matches_oracle_state_after_max_iters(gpu, orc, variant, depth, n, max_iters):
    """Same cut sequence as the oracle loop (cursor positions, iteration count, status) and the same state.
    The filter-design problem amplifies rounding: ONE ulp on one entry of Q drifts the oracle's own run by
    ~2e-11 after 64 cuts and ~5e-10 after 300 (cond(Q) ~ 1e7-1e8), so beyond 16 cuts the state tolerance is
    100 x that measured drift (never below 1e-10); up to 16 cuts it is the plain 1e-10."""
    c = CONSTANT_SETS["corrected"]
    dev_o = gpu.LowpassOracle(n, *c)
    g, _ = make_spaces(gpu, orc, variant, n, 40.0, depth)
    cpu_o, o, xbo, nitero, gammao, last = cpu_lockstep(orc, variant, n, c, 40.0, max_iters, False)
    xb, niter, gamma = dev_o.cutting_plane_optim(g, c[4], max_iters, 1e-14)
    assert niter == nitero
    if variant == "ell":
        assert niter == max_iters and last == 0
    else:
        assert last in (0, orc.NOSOLN)  # the reference's EllStable loses the factorisation on this problem (F5)
    assert (xb is None) == (xbo is None)
    sd, sc = dev_o.state(), cpu_o.state()
    assert (sd["idx1"], sd["idx2"], sd["idx3"]) == (sc["idx1"], sc["idx2"], sc["idx3"])
    tol = 1e-10 if max_iters <= 16 else max(1e-10, 100.0 * sensitivity(orc, variant, n, c, 40.0, max_iters))
    assert tol < 1e-5, tol
    assert abs(gamma - gammao) <= tol * abs(gammao)
    if xbo is not None:
        assert rel_inf(xb, xbo) <= tol
    assert_state_close(g, o, tol=tol, what=f"n={n} after {max_iters} iterations")


@pytest.mark.parametrize("variant,depth", [("ell", 1), ("ell", 8), ("stable", 1)])
def test_optim_loop_stops_on_tolerance_with_the_last_update_applied(gpu, orc, variant, depth):
    n = 24
    c = CONSTANT_SETS["corrected"]
    dev_o = gpu.LowpassOracle(n, *c)
    cpu_o = orc.OracleLowpass(n, *c)
    g, o = make_spaces(gpu, orc, variant, n, 40.0, depth)
    # find a tolerance that the oracle run crosses (a new record low of tsq) after ten or more iterations
    probe_o = orc.OracleLowpass(n, *c)
    probe_s = make_spaces(gpu, orc, variant, n, 40.0, depth)[1]
    tsqs = []
    gamma = c[4]
    for it in range(60):
        (gr, (b0, b1)), sh, gamma = probe_o.assess_optim(np.array(probe_s.xc), gamma)
        if probe_s.update(1 if sh else 0, gr, b0, b1) != 0:
            break
        tsqs.append(probe_s.tsq)
    lows = [i for i in range(10, len(tsqs)) if tsqs[i] < 0.98 * min(tsqs[:i])]
    assert lows, tsqs
    stop_at = lows[0]
    tol = 0.5 * (tsqs[stop_at] + min(tsqs[:stop_at]))
    assert tsqs[stop_at] < tol <= min(tsqs[:stop_at])
    xb, niter, gamma_d = dev_o.cutting_plane_optim(g, c[4], 5000, tol)
    xbo, nitero, gamma_c, last = cpu_o.cutting_plane_optim(o, c[4], 5000, tol)
    assert niter == nitero == stop_at and last == 0
    assert abs(gamma_d - gamma_c) <= 1e-10 * abs(gamma_c)
    assert_state_close(g, o, tol=1e-8, what="tolerance stop")  # includes the shrink of the stopping update
    # and the loop can be resumed: same continuation on both sides
    xb2, niter2, gamma_d2 = dev_o.cutting_plane_optim(g, gamma_d, 20, 0.0)
    xbo2, nitero2, gamma_c2, _ = cpu_o.cutting_plane_optim(o, gamma_c, 20, 0.0)
    assert niter2 == nitero2
    assert_state_close(g, o, tol=1e-8, what="resumed")


@pytest.mark.parametrize("variant,depth", [("ell", 1), ("ell", 8)])
@pytest.mark.parametrize("n", [32, 48])
def test_optim_loop_full_run(gpu, orc, variant, depth, n):
    """Whole corrected run to its natural end (thousands of cuts).  Rounding differences between the butterfly
    and the left-fold dot products are amplified over such a run, so the end point is compared loosely; the
    exact-sequence comparison is the max_iters test above."""
    c = CONSTANT_SETS["corrected"]
    dev_o = gpu.LowpassOracle(n, *c)
    cpu_o = orc.OracleLowpass(n, *c)
    g, o = make_spaces(gpu, orc, variant, n, 40.0, depth)
    xb, niter, gamma = dev_o.cutting_plane_optim(g, c[4], 50000, 1e-14)
    xbo, nitero, gammao, last = cpu_o.cutting_plane_optim(o, c[4], 50000, 1e-14)
    assert xb is not None and xbo is not None
    assert abs(niter - nitero) <= 0.05 * nitero + 5, (niter, nitero)
    assert abs(gamma - gammao) <= 1e-3 * gammao, (gamma, gammao)
    # x_best really is feasible for the returned gamma
    vals = cpu_o.spectrum @ xb
    st = cpu_o.state()
    assert np.all(vals[:st["nwpass"]] <= c[3] + 1e-9) and np.all(vals[:st["nwpass"]] >= c[2] - 1e-9)
    assert np.all(vals[st["nwstop"]:] <= gamma + 1e-9) and np.all(vals >= -1e-9)


@pytest.mark.parametrize("variant,depth", [("ell", 1), ("ell", 8), ("stable", 1)])
def test_feas_loop(gpu, orc, variant, depth):
    n = 32
    c = CONSTANT_SETS["loose"]
    dev_o = gpu.LowpassOracle(n, *c)
    cpu_o = orc.OracleLowpass(n, *c)
    g, o = make_spaces(gpu, orc, variant, n, 40.0, depth)
    x, niter = dev_o.cutting_plane_feas(g, 2000, 1e-14)
    xo, nitero, last = cpu_o.cutting_plane_feas(o, 2000, 1e-14)
    assert (x is None) == (xo is None) and niter == nitero
    if variant == "ell":
        assert xo is not None and niter > 3
    if xo is not None:
        assert rel_inf(x, xo) <= 1e-8
    assert rel_inf(g.xc(), np.array(o.xc)) <= 1e-8
    # infeasible band edges never produce a point: both sides stop on the same iteration and status
    c2 = CONSTANT_SETS["as_written"]
    dev_o, cpu_o = gpu.LowpassOracle(n, *c2), orc.OracleLowpass(n, *c2)
    g, o = make_spaces(gpu, orc, variant, n, 40.0, depth)
    x, niter = dev_o.cutting_plane_feas(g, 2000, 1e-14)
    xo, nitero, last = cpu_o.cutting_plane_feas(o, 2000, 1e-14)
    assert x is None and xo is None and niter == nitero == 0 and last == orc.NOSOLN


def test_loop_argument_checks(gpu):
    dev_o = gpu.create_lowpass_case(16)
    g = gpu.Ell.new_with_scalar(1.0, np.zeros(8))
    with pytest.raises(gpu.capi.EllHipError):
        dev_o.cutting_plane_optim(g, 0.1, 10, 1e-8)
    g = gpu.Ell.new_with_scalar(1.0, np.zeros(16))
    xb, niter, gamma = dev_o.cutting_plane_optim(g, 0.1, 0, 1e-8)
    assert xb is None and niter == 0 and gamma == 0.1


# ---- through the C++ host mirror (ellalgo-rs_amd/host/ellhip/lowpass_hip.hpp) ---------------------------

def test_cpp_host_mirror_three_drivers_agree_with_the_oracle_loop(gpu, orc):
    import cpp_build
    exe = cpp_build.build_runner("lowpass_runner.cpp", "hip")
    res = cpp_build.run_json_lines(exe)
    _, _, lp_sq, up_sq, sp_sq = orc.lowpass_case(False)
    # the reference's own runs: (None, 0) with gamma untouched
    for case, g0 in [("run_lowpass_host", sp_sq), ("run_lowpass_device", sp_sq), ("stress_high_dimension_device", sp_sq),
                     ("stress_many_iterations_device", 1e-12)]:
        r = res[case]
        assert r["niter"] == 0 and not r["has_x"] and r["gamma"] == g0, case
    r = res["oracle_zero"]  # assess_feas(0): row 0 negated, ParallelCut(lp_sq, Some(up_sq))
    assert r["has_x"] and r["gamma"] == lp_sq and r["tsq"] == up_sq
    assert r["x"] == [-1.0] + [-2.0] * 31
    assert res["oracle_direct"]["has_x"] and math.isfinite(res["oracle_direct"]["gamma"])
    # corrected constants: 200 iterations through the three drivers vs the oracle loop
    c = CONSTANT_SETS["corrected"]
    cpu_o, o, xbo, nitero, gammao, last = cpu_lockstep(orc, "ell", 32, c, 40.0, 200, False)
    tol = max(1e-10, 100.0 * sensitivity(orc, "ell", 32, c, 40.0, 200))
    for case in ("corrected_host", "corrected_pipelined", "corrected_device"):
        r = res[case]
        assert r["niter"] == nitero == 200 and r["has_x"] == (xbo is not None), case
        assert abs(r["gamma"] - gammao) <= tol * gammao, case
        assert abs(r["tsq"] - o.tsq) <= tol * o.tsq, case
        if xbo is not None:
            assert rel_inf(np.array(r["x"]), xbo) <= tol, case
    # the three drivers issue the same arithmetic on the device: identical bits
    assert res["corrected_host"]["x"] == res["corrected_device"]["x"]
    assert res["corrected_host"]["tsq"] == res["corrected_device"]["tsq"]
    assert res["corrected_pipelined"]["tsq"] == res["corrected_device"]["tsq"]
    # feasibility loop
    cl = CONSTANT_SETS["loose"]
    fo = orc.OracleLowpass(32, *cl)
    fe = orc.OracleEll.new_with_scalar(40.0, np.zeros(32))
    xo, nitero, last = fo.cutting_plane_feas(fe, 2000, 1e-14)
    for case in ("feas_host", "feas_device"):
        r = res[case]
        assert r["niter"] == nitero and r["has_x"] == (xo is not None), case
        assert rel_inf(np.array(r["x"]), xo) <= 1e-8
    assert res["feas_host"]["x"] == res["feas_device"]["x"]


@pytest.mark.parametrize("depth", [8, 16])
@pytest.mark.parametrize("max_iters", [16, 40, 70])
def test_optim_loop_on_the_lower_triangle_schedule(gpu, orc, depth, max_iters, monkeypatch):
    """The device-resident loop over the schedule a large handle runs by default (lower-triangle GEMV, recorded
    updates applied 8 / 16 at a time; forced on at n = 640 here): same cut sequence and state as the oracle loop."""
    set_default("SYMV_MIN_N", 512)
    n = 640
    c = CONSTANT_SETS["corrected"]
    dev_o = gpu.LowpassOracle(n, *c)
    g, _ = make_spaces(gpu, orc, "ell", n, 40.0, depth)
    assert g.defer_depth == depth
    cpu_o, o, xbo, nitero, gammao, last = cpu_lockstep(orc, "ell", n, c, 40.0, max_iters, False)
    xb, niter, gamma = dev_o.cutting_plane_optim(g, c[4], max_iters, 1e-14)
    assert niter == nitero == max_iters and last == 0
    assert (xb is None) == (xbo is None)
    sd, sc = dev_o.state(), cpu_o.state()
    assert (sd["idx1"], sd["idx2"], sd["idx3"]) == (sc["idx1"], sc["idx2"], sc["idx3"])
    tol = 1e-10 if max_iters <= 16 else max(1e-10, 100.0 * sensitivity(orc, "ell", n, c, 40.0, max_iters))
    assert tol < 1e-5, tol
    assert abs(gamma - gammao) <= tol * abs(gammao)
    if xbo is not None:
        assert rel_inf(xb, xbo) <= tol
    assert_state_close(g, o, tol=tol, what=f"lower-triangle schedule depth {depth}, {max_iters} iterations")
    # resume (the loop leaves recorded updates behind; the next call continues from them)
    xb2, niter2, gamma2 = dev_o.cutting_plane_optim(g, gamma, 10, 1e-14)
    xbo2, nitero2, gammao2, _ = cpu_o.cutting_plane_optim(o, gammao, 10, 1e-14)
    assert niter2 == nitero2
    assert_state_close(g, o, tol=max(tol, 1e-9), what="resumed")
