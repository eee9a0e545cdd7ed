"""EllStable on a NON-trivial factor (round-1 verdict, item 1).

`EllStable::new_with_matrix` is public and accepts any packed buffer (src/ell_stable.rs:18-27).  From the identity /
a diagonal (`new_with_scalar`, `new`) the bug-compatible arithmetic keeps the strict upper triangle U and the scratch
triangle S exactly zero forever, so tests that start there compare the triangular-solve kernels on
0 * x = 0 only.  Everything here starts from a random unit-upper-triangular factor, a random positive diagonal and
junk in the scratch triangle, so the panel indexing of the forward / backward solves, the transposed scratch stores and
the tile transpose of the factor update (src/ell_stable.rs:61-69, :93-98, :107-119) are compared on real data: the WHOLE
buffer (D, U, S), xc, kappa and tsq against the CPU oracle to the north-star tolerance 1e-10.
"""
import numpy as np
import pytest

from util import TOL, assert_state_close, random_factor, rel_inf, run_mixed_stable, set_default

pytestmark = pytest.mark.gpu
DEFAULT_SOLVE = 3   # what a new EllStable handle starts with (ELLHIP_OPT_STABLE_SOLVE)

# block edges of the 64-wide halves and the 128-wide blocks of the solves, ragged last blocks, several blocks
SIZES = [2, 3, 63, 64, 65, 127, 128, 129, 257, 1000, 2048]


def _pair(gpu, orc, n, seed, kappa=1.5):
    f = random_factor(n, seed)
    xc0 = np.linspace(-1.0, 1.0, n)
    return (gpu.EllStable.new_with_matrix(kappa, f, xc0), orc.OracleEllStable.new_with_matrix(kappa, f, xc0), f)


def _offdiag_nonzeros(m):
    return int(np.count_nonzero(m - np.diag(np.diag(m))))


@pytest.mark.parametrize("solve,every", [(2, 4), (3, 4), (3, 0)])
@pytest.mark.parametrize("n", SIZES)
def test_mixed_sequence_on_random_factor_matches_oracle(gpu, orc, n, solve, every):
    """All six EllCalc entry points incl. a failing cut every 8th step (it rewrites the scratch triangle only).
    solve = 3: the mirrored layout (no scratch triangle, the factor update applied by the next solves); every = 4: the buffer
    is observed every 4th cut (the layout is left -- scratch triangle rebuilt, pending factor update applied -- and entered
    again), every = 0: only at the end, after up to 24 cuts inside the layout."""
    set_default("STABLE_SOLVE", solve)
    g, o, f = _pair(gpu, orc, n, 9000 + n)
    assert g.get_option(gpu.capi.OPT_STABLE_SOLVE) == solve
    k = 24 if n <= 1000 else 16
    nsucc = run_mixed_stable(g, o, k, seed=700 + n, check_every=every)
    assert nsucc >= k // 2
    assert_state_close(g, o, what=f"n={n} final")
    m = g.mq
    if n > 2:
        # the comparison is not a comparison of zeros: factor AND scratch triangle are populated
        assert np.count_nonzero(np.triu(m, 1)) >= (n * (n - 1) // 2) * 9 // 10
        assert np.count_nonzero(np.tril(m, -1)) >= (n * (n - 1) // 2) * 9 // 10
    # each triangle on its own scale (the scratch products are ~0.1/sqrt(n) times smaller than the diagonal)
    mo = o.mq
    assert rel_inf(np.triu(m, 1), np.triu(mo, 1)) <= TOL
    assert rel_inf(np.tril(m, -1), np.tril(mo, -1)) <= TOL
    assert rel_inf(np.diag(m), np.diag(mo)) <= TOL


@pytest.mark.parametrize("n", [2, 65, 130, 257, 1000])
def test_first_update_overwrites_all_scratch_junk(gpu, orc, n):
    """src/ell_stable.rs:61-69: the forward solve writes S[i][j] for every j < i before anything reads it, so the
    caller's junk in the strict lower triangle must not survive the first cut -- not even a failing one."""
    f = random_factor(n, 31 + n)
    f_clean = np.triu(f)
    rng = np.random.default_rng(n)
    gr = rng.standard_normal(n)
    for beta, want in ((0.01, 0), (1e6, 1)):
        a = gpu.EllStable.new_with_matrix(1.0, f, np.zeros(n))
        b = gpu.EllStable.new_with_matrix(1.0, f_clean, np.zeros(n))
        o = orc.OracleEllStable.new_with_matrix(1.0, f, np.zeros(n))
        assert int(a.update_bias_cut((gr, beta))) == int(b.update_bias_cut((gr, beta))) == o.update(0, gr, beta) == want
        assert np.array_equal(a.mq, b.mq) and np.array_equal(a.xc(), b.xc()) and a.kappa == b.kappa
        assert_state_close(a, o, what=f"n={n} beta={beta}")
        if want == 1:   # failed cut: diagonal and factor untouched, scratch rewritten (src/ell_stable.rs:66,88-90)
            assert np.array_equal(np.triu(a.mq), f_clean)


@pytest.mark.parametrize("n", [65, 129, 257, 1000, 2048])
def test_persistent_equals_per_block_launches_on_random_factor(gpu, n):
    """The flag-chained single-launch solves against one launch per block (no inter-workgroup hand-off): same
    arithmetic in the same order, so the same bits -- now on data where an indexing slip would show."""
    f = random_factor(n, 77 + n)
    set_default("STABLE_SOLVE", 2)
    a = gpu.EllStable.new_with_matrix(1.0, f, np.zeros(n))
    set_default("STABLE_SOLVE", 0)
    b = gpu.EllStable.new_with_matrix(1.0, f, np.zeros(n))
    set_default("STABLE_SOLVE", 2)
    rng = np.random.default_rng(5 * n)
    for i in range(10):
        gr = rng.standard_normal(n)
        gr /= np.linalg.norm(gr)
        beta = 5.0 if i == 6 else 0.05 * rng.random()   # cut 6 fails
        sa, sb = int(a.update_bias_cut((gr, beta))), int(b.update_bias_cut((gr, beta)))
        assert sa == sb == (1 if i == 6 else 0)
        assert a.tsq() == b.tsq() and a.kappa == b.kappa
    assert np.array_equal(a.xc(), b.xc())
    assert np.array_equal(a.mq, b.mq)
    assert _offdiag_nonzeros(a.mq) > 0


def _same_to_rounding(x, ref, tol=1e-12, what=""):
    """The mirrored layout against an eager form: every part of the state on its own scale, far inside the parity tolerance."""
    mx, mr = x.mq, ref.mq
    assert rel_inf(np.triu(mx, 1), np.triu(mr, 1)) <= tol, f"{what}: factor"
    assert rel_inf(np.tril(mx, -1), np.tril(mr, -1)) <= tol, f"{what}: scratch triangle"
    assert rel_inf(np.diag(mx), np.diag(mr)) <= tol, f"{what}: diagonal"
    assert rel_inf(x.xc(), ref.xc()) <= tol and abs(x.kappa - ref.kappa) <= tol * abs(ref.kappa), f"{what}: xc / kappa"
    assert abs(x.tsq() - ref.tsq()) <= tol * abs(ref.tsq()), f"{what}: tsq"


@pytest.mark.parametrize("n", [2, 63, 64, 65, 129, 300, 513, 1000, 2048, 2048 + 128, 2049, 8192, 8200, 8191])
def test_default_solves_and_factor_update_equal_the_plain_kernels(gpu, n):
    """Every EAGER EllStable kernel form must give the bits of the plain path (one launch per block, factor update reading the
    scratch triangle through LDS transposes): both solves with a helper workgroup per block (k_st_fwd_helped,
    k_st_bwd_factor_helped) and the factor update computed from U alone inside the backward solve's launch (the scratch entry
    it would read IS fl(U * w)), the persistent solves without helpers with the row-wise factor kernel beside them
    (k_st_fwd_persist, k_st_bwd_persist, k_st_factor_rows), and the forms switched on an existing handle with
    ellhip_set_option.  The MIRRORED layout (STABLE_SOLVE = 3, the default: no scratch triangle inside the loop, the factor
    update kept as one running scale per row) rounds once per use where the eager forms round twice per update: it must agree
    with them to 1e-12 on every part of the state, also when it is entered and left between cuts, observed right after a
    failing cut, or mixed with the eager forms on one handle.  Odd and even n, ragged last blocks, one block, failing cuts in
    the middle; 8191 / 8192 / 8200 straddle the size where the factor tiles switch from 512- to 2048-column segments and the
    chain workgroups stop pulling tiles before their turn."""
    capi = gpu.capi
    f = random_factor(n, 271 + n)
    a = gpu.EllStable.new_with_matrix(1.0, f, np.zeros(n))     # the default: the mirrored layout throughout, never observed
    assert a.get_option(capi.OPT_STABLE_SOLVE) == DEFAULT_SOLVE and a.get_option(capi.OPT_STABLE_FACTOR) == 2
    set_default("STABLE_SOLVE", 0)
    set_default("STABLE_FACTOR", 0)
    b = gpu.EllStable.new_with_matrix(1.0, f, np.zeros(n))
    assert b.get_option(capi.OPT_STABLE_SOLVE) == 0 and b.get_option(capi.OPT_STABLE_FACTOR) == 0
    set_default("STABLE_SOLVE", 1)
    set_default("STABLE_FACTOR", 1)
    c = gpu.EllStable.new_with_matrix(1.0, f, np.zeros(n))
    set_default("STABLE_SOLVE", 2)
    set_default("STABLE_FACTOR", 2)
    d = gpu.EllStable.new_with_matrix(1.0, f, np.zeros(n))   # walks through the eager forms, one per cut
    e2 = gpu.EllStable.new_with_matrix(1.0, f, np.zeros(n))  # helped solves + pulled factor tiles throughout
    set_default("STABLE_SOLVE", 3)
    e3 = gpu.EllStable.new_with_matrix(1.0, f, np.zeros(n))  # the mirrored layout, observed in the middle
    d3 = gpu.EllStable.new_with_matrix(1.0, f, np.zeros(n))  # walks through all forms, the mirrored layout among them
    forms = [(0, 0), (1, 0), (1, 1), (2, 1), (2, 2), (0, 1), (2, 0), (1, 2), (2, 2), (0, 0), (1, 1), (2, 0), (1, 2), (2, 1)]
    forms3 = [(0, 0), (1, 0), (3, 1), (3, 1), (3, 2), (3, 0), (2, 1), (2, 2), (3, 2), (0, 1), (3, 0), (3, 0), (1, 2), (3, 1)]
    fails = (5, 10)          # (in forms3 both inside a stretch of the mirrored layout: a failed cut's scratch triangle is lazy too)
    rng = np.random.default_rng(13 * n)
    for i in range(len(forms)):
        gr = rng.standard_normal(n)
        gr /= np.linalg.norm(gr)
        beta = 5.0 if i in fails else 0.05 * rng.random()
        for h, fm in ((d, forms[i]), (d3, forms3[i])):
            h.set_option(capi.OPT_STABLE_SOLVE, fm[0])
            h.set_option(capi.OPT_STABLE_FACTOR, fm[1])
        stats = [int(x.update_bias_cut((gr, beta))) for x in (a, b, c, d, e2, e3, d3)]
        assert stats == [1 if i in fails else 0] * 7, (i, stats)
        assert b.tsq() == c.tsq() == d.tsq() == e2.tsq() and b.kappa == c.kappa == d.kappa == e2.kappa
        for x in (a, e3, d3):
            assert abs(x.tsq() - b.tsq()) <= 1e-12 * abs(b.tsq()) and abs(x.kappa - b.kappa) <= 1e-12 * abs(b.kappa)
        if i in (3, 5, 8):   # the mirrored handle observed after a success, right after a failing cut, and again
            _same_to_rounding(e3, b, what=f"mirrored layout, buffer after cut {i}")
    qb = b.mq
    for x in (c, d, e2):
        assert np.array_equal(x.xc(), b.xc()) and np.array_equal(x.mq, qb)
    for name, x in (("default", a), ("observed", e3), ("mixed forms", d3)):
        _same_to_rounding(x, b, what=f"mirrored layout ({name}) at the end")
    with pytest.raises(capi.EllHipError):
        gpu.Ell.new_with_scalar(1.0, np.zeros(8)).set_option(capi.OPT_STABLE_SOLVE, 1)   # an EllStable option
    with pytest.raises(capi.EllHipError):
        a.set_option(capi.OPT_SYMV, 0)                                                   # an Ell option
    with pytest.raises(capi.EllHipError):
        a.set_option(capi.OPT_STABLE_SOLVE, 4)


def test_failed_cut_in_the_middle_of_a_sequence(gpu, orc):
    n = 200
    g, o, f = _pair(gpu, orc, n, 4242)
    rng = np.random.default_rng(11)
    for i in range(9):
        gr = rng.standard_normal(n)
        if i == 4:
            k0, x0, up0 = g.kappa, g.xc(), np.triu(g.mq)
            assert int(g.update_bias_cut((gr, 1e6))) == o.update(0, gr, 1e6) == 1
            assert g.kappa == k0 and np.array_equal(g.xc(), x0) and np.array_equal(np.triu(g.mq), up0)
        else:
            assert int(g.update_central_cut((gr, 0.0))) == o.update(1, gr, 0.0) == 0
        assert_state_close(g, o, what=f"cut {i}")


@pytest.mark.parametrize("solve", [2, 3])
def test_clone_and_queue_on_random_factor(gpu, orc, solve):
    from ellalgo_rs_amd import synth
    set_default("STABLE_SOLVE", solve)
    n, k = 320, 10
    kinds, grads, b0, b1 = synth.deep_cuts(n, k)
    f = random_factor(n, 99)
    a = gpu.EllStable.new_with_matrix(1.0, f, np.zeros(n))
    a.queue_upload(kinds, grads, b0, b1)
    a.queue_run(0, k)
    st, ts = a.queue_results()
    assert list(st) == [0] * k
    b = gpu.EllStable.new_with_matrix(1.0, f, np.zeros(n))
    o = orc.OracleEllStable.new_with_matrix(1.0, f, np.zeros(n))
    for i in range(5):
        b.update_bias_cut((grads[i], b0[i]))
    c = b.clone()
    for i in range(5, k):
        b.update_bias_cut((grads[i], b0[i]))
        c.update_bias_cut((grads[i], b0[i]))
    for i in range(k):
        assert o.update(0, grads[i], b0[i]) == 0
        assert abs(ts[i] - o.tsq) <= TOL * abs(o.tsq)
    if solve == 3:
        # taking the clone observed b's buffer (the mirrored layout was left and entered again: U = fl(U_base r) rounds once
        # more), so the three histories agree to rounding, not to the bit
        _same_to_rounding(b, a, what="direct + clone vs queue")
        _same_to_rounding(c, a, what="clone vs queue")
    else:
        assert np.array_equal(a.mq, b.mq) and np.array_equal(b.mq, c.mq)
        assert np.array_equal(a.xc(), b.xc()) and a.kappa == b.kappa == c.kappa
    assert_state_close(a, o, what="queue vs oracle")
    assert_state_close(c, o, what="clone vs oracle")


def test_synth_stable_factor_long_run(gpu, orc):
    """The factor bench.py's EllStable workloads start from (synth.stable_factor), 120 deep cuts at n = 1024."""
    from ellalgo_rs_amd import synth
    n, k = 1024, 120
    f = synth.stable_factor(n)
    kinds, grads, b0, b1 = synth.deep_cuts(n, k)
    g = gpu.EllStable.new_with_matrix(1.0, f, np.zeros(n))
    o = orc.OracleEllStable.new_with_matrix(1.0, f, np.zeros(n))
    g.queue_upload(kinds, grads, b0, b1)
    g.queue_run(0, k)
    st, ts = g.queue_results()
    for i in range(k):
        assert o.update(0, grads[i], b0[i]) == 0
        assert abs(ts[i] - o.tsq) <= TOL * abs(o.tsq), i
    assert np.all(st == 0)
    assert_state_close(g, o, what="120 cuts")


@pytest.mark.parametrize("n", [200, 1000, 2176])
def test_halted_queue_in_the_mirrored_layout(gpu, orc, n):
    """A queue whose cut 7 fails, on the mirrored layout: the forward solve of the failing cut has already applied the factor
    update of cut 6 to U (in place) and produced the w the failed cut's scratch triangle is made of; every kernel behind it is
    a no-op; reading the results leaves the layout (scratch triangle of cut 7, nothing pending on U); a second run on the
    still-halted queue must not touch the buffer; direct updates afterwards enter the layout again."""
    from ellalgo_rs_amd import synth
    set_default("STABLE_SOLVE", 3)
    k, bad = 14, 7
    kinds, grads, b0, b1 = synth.deep_cuts(n, k)
    b0 = b0.copy()
    b0[bad] = 1e6
    f = random_factor(n, 555 + n)
    g = gpu.EllStable.new_with_matrix(1.0, f, np.zeros(n))
    o = orc.OracleEllStable.new_with_matrix(1.0, f, np.zeros(n))
    g.queue_upload(kinds, grads, b0, b1)
    g.queue_run(0, 10)
    g.queue_run(10, 4)          # halted: nothing runs
    st, ts = g.queue_results()
    for i in range(bad + 1):
        assert o.update(0, grads[i], b0[i]) == (1 if i == bad else 0)
        assert abs(ts[i] - o.tsq) <= TOL * abs(o.tsq)
    assert list(st[:bad]) == [0] * bad and int(st[bad]) == 1 and all(int(x) == 3 for x in st[bad + 1:])
    assert_state_close(g, o, what="after the halt")
    assert rel_inf(np.tril(g.mq, -1), np.tril(o.mq, -1)) <= TOL      # the failing cut's scratch triangle
    for i in range(bad + 1, k):
        b0[i] = 0.02
        assert int(g.update_bias_cut((grads[i], b0[i]))) == o.update(0, grads[i], b0[i]) == 0
    assert_state_close(g, o, what="direct updates after the halt")
