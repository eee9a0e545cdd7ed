"""GPU: LDLTMgr / LMIOracle / LMI0Oracle on the device (include/ellhip_lmi.h) against the CPU oracle and the
reference's own known answers, through the C ABI.  The factor (`storage` on every row the reference touches),
`pos` and ep are compared EXACTLY; the witness and the cut gradient to rounding."""

import numpy as np
import pytest

from test_lmi_oracle_cpu import (B1, B2, CHOL1, CHOL2, CHOL3, CHOL8, F1, F2, LMI0_F, MyLmiOracle, run_lmi)

pytestmark = pytest.mark.gpu


def random_pencil(rng, n, m, shift):
    """B positive definite by `shift`, F_k symmetric: F(x) goes indefinite as |x| grows"""
    a = rng.standard_normal((m, m))
    B = a @ a.T / m + shift * np.eye(m)
    F = rng.standard_normal((n, m, m))
    F = (F + F.transpose(0, 2, 1)) / 2.0
    return F, B


def compare_factor(dev, cpu_ldlt, m):
    assert dev.pos == cpu_ldlt.pos
    p = cpu_ldlt.pos[1]
    rows = m if p == 0 else p  # the reference stops at the failing row; later rows are untouched there
    np.testing.assert_array_equal(dev.storage[:rows, :rows], cpu_ldlt.storage[:rows, :rows])


def test_reference_known_answers(gpu):
    m = gpu.LDLTMgr(3)
    assert m.factorize(CHOL1) and m.is_spd()                       # ldlt_mgr.rs:185-190
    m4 = gpu.LDLTMgr(4)
    assert not m4.factorize(CHOL2) and m4.pos == (0, 2)            # :192-199
    m3 = gpu.LDLTMgr(3)
    assert not m3.factorize(CHOL3)                                  # :201-209
    assert m3.pos == (0, 1) and m3.wit[0] == 1.0 and m3.witness() == 0.0
    assert not gpu.LDLTMgr(3).factorize(CHOL8)                      # :243-248
    mat = np.array([[1.0, 0.5, 0.5], [0.5, 1.25, 0.75], [0.5, 0.75, 1.5]])
    ms = gpu.LDLTMgr(3)
    assert ms.factorize(mat)
    assert np.allclose(ms.sqrt(), [[1.0, 0.5, 0.5], [0.0, 1.0, 0.5], [0.0, 0.0, 1.0]], rtol=0, atol=1e-15)  # :257-268
    assert gpu.LMIOracle(F1, B1).assess_feas(np.zeros(3)) is None  # tests/lmi_tests.rs:58-63
    assert gpu.LMI0Oracle(F1).assess_feas(np.zeros(3)) is not None  # :65-71
    assert gpu.LMI0Oracle(LMI0_F).assess_feas(np.array([1.0, 0.0, 1.0])) is None   # :85-91
    g, ep = gpu.LMI0Oracle(LMI0_F).assess_feas(np.array([-1.0, 0.0, -1.0]))         # :93-104
    assert g[0] == pytest.approx(-1.0) and abs(g[1]) < 1e-12 and abs(g[2]) < 1e-12 and ep == pytest.approx(1.0)
    assert gpu.LMI0Oracle(LMI0_F).assess_feas(np.array([1.0, 1.0, 1.0])) is not None  # :106-112


@pytest.mark.parametrize("m", [1, 2, 3, 7, 31, 32, 33, 64, 65, 100, 257, 600])
def test_factorize_matches_oracle_exactly(gpu, orc, m):
    rng = np.random.default_rng(m)
    a = rng.standard_normal((m, m))
    spd = a @ a.T + m * np.eye(m)
    dev, cpu = gpu.LDLTMgr(m), orc.OracleLDLT(m)
    assert dev.factorize(spd) and cpu.factorize(spd)
    compare_factor(dev, cpu, m)
    np.testing.assert_array_equal(dev.sqrt(), cpu.sqrt())
    # a failing pivot at several depths
    for frac in (0.0, 0.3, 0.7, 1.0):
        k = min(m - 1, int(frac * m))
        bad = spd.copy()
        bad[k, k] -= 2.0 * np.linalg.eigvalsh(spd)[-1] + 1.0
        assert dev.factorize(bad) == cpu.factorize(bad) is False
        compare_factor(dev, cpu, m)
        ep_c = cpu.witness()
        assert dev.witness() == ep_c
        s, e = cpu.pos
        assert np.allclose(dev.wit[s:e], cpu.wit[s:e], rtol=1e-11, atol=1e-13 * np.max(np.abs(cpu.wit[s:e])))
        assert np.all(dev.wit[e:] == 0.0)


@pytest.mark.parametrize("n,m", [(3, 2), (3, 3), (5, 17), (16, 64), (40, 100), (24, 300), (8, 700)])
def test_lmi_oracle_matches_cpu(gpu, orc, n, m):
    rng = np.random.default_rng(100 * n + m)
    F, B = random_pencil(rng, n, m, shift=1.0)
    dev, cpu = gpu.LMIOracle(F, B), orc.OracleLMI(F, B)
    dev0, cpu0 = gpu.LMI0Oracle(F), orc.OracleLMI(F)
    ncut = nfeas = 0
    for it in range(12):
        x = rng.standard_normal(n) * [0.0, 0.01, 0.05, 0.2, 1.0][it % 5]
        for d, c, is0 in ((dev, cpu, False), (dev0, cpu0, True)):
            rd, rc = d.assess_feas(x), c.assess_feas(x)
            assert (rd is None) == (rc is None), (it, is0)
            compare_factor(d, c.ldlt, m)
            if rc is None:
                nfeas += 1
                continue
            ncut += 1
            gd, epd = rd[0], (rd[1] if is0 else rd[1].beta)
            gc, epc = rc
            assert epd == epc
            scale = np.max(np.abs(gc)) + 1e-300
            assert np.max(np.abs(gd - gc)) <= 1e-10 * scale, (it, is0)
    assert ncut > 0 and nfeas > 0


def test_lmi_runs_of_the_reference_through_the_device_oracle(gpu, orc):
    """tests/lmi_tests.rs:199-217: x_best is Some, < 300 iterations (Ell) / < 400 (EllStable); and the very same
    iteration count as the CPU oracle pair."""
    make_dev = lambda f, b: _DevLmiAdapter(gpu.LMIOracle(f, b))
    for space_cls, limit in ((orc.OracleEll, 300), (orc.OracleEllStable, 400)):
        xb_c, n_c = run_lmi(space_cls.new_with_scalar(10.0, np.zeros(3)), MyLmiOracle(orc.OracleLMI))
        xb_d, n_d = run_lmi(space_cls.new_with_scalar(10.0, np.zeros(3)), MyLmiOracle(make_dev))
        assert xb_d is not None and n_d < limit
        assert n_d == n_c and np.allclose(xb_d, xb_c, rtol=1e-9, atol=1e-12)


class _DevLmiAdapter:
    def __init__(self, o):
        self.o = o

    def assess_feas(self, x):
        r = self.o.assess_feas(x)
        return None if r is None else (r[0], r[1].beta)


def test_large_block(gpu, orc):
    """m = 1500: 47 panels, forming in 6 lazy slabs; exact agreement of the decision and of the factor."""
    n, m = 4, 1500
    rng = np.random.default_rng(9)
    F, B = random_pencil(rng, n, m, shift=0.5)
    dev, cpu = gpu.LMIOracle(F, B), orc.OracleLMI(F, B)
    assert dev.assess_feas(np.zeros(n)) is None and cpu.assess_feas(np.zeros(n)) is None
    compare_factor(dev, cpu.ldlt, m)
    x = np.array([0.3, -0.2, 0.1, 0.25])
    rd, rc = dev.assess_feas(x), cpu.assess_feas(x)
    assert rd is not None and rc is not None
    compare_factor(dev, cpu.ldlt, m)
    assert rd[1].beta == rc[1]
    assert np.max(np.abs(rd[0] - rc[0])) <= 1e-10 * np.max(np.abs(rc[0]))


def test_lmi_argument_checks(gpu):
    import ctypes as C
    h = C.c_void_p()
    one = (C.c_double * 1)(1.0)
    assert gpu.capi.load().ellhip_lmi_create(C.byref(h), 0, 8193, None, one, -1) == gpu.capi.E_INVALID  # m > 8192
    with pytest.raises(ValueError):
        gpu.LMIOracle(np.zeros((2, 3, 4)), np.eye(3))
    o = gpu.LMIOracle(F1, B1)
    with pytest.raises(ValueError):
        o.assess_feas(np.zeros(4))
    m = gpu.LDLTMgr(3)
    assert not m.factorize(CHOL3)
    with pytest.raises(AssertionError):
        m.sqrt()


def test_cpp_mirror_runs_the_reference_lmi_problem_on_device_space_and_device_oracle(gpu, orc):
    import cpp_build
    exe = cpp_build.build_runner("lmi_runner.cpp", "hip")
    res = cpp_build.run_json_lines(exe)
    assert res["chol1"]["has_x"]
    # the same loops on the CPU oracle pair
    xb_e, n_e = run_lmi(orc.OracleEll.new_with_scalar(10.0, np.zeros(3)), MyLmiOracle(orc.OracleLMI))
    xb_s, n_s = run_lmi(orc.OracleEllStable.new_with_scalar(10.0, np.zeros(3)), MyLmiOracle(orc.OracleLMI))
    r = res["lmi_lazy"]
    assert r["has_x"] and r["niter"] < 300 and abs(r["niter"] - n_e) <= 2          # tests/lmi_tests.rs:199-205
    assert np.allclose(r["x"], xb_e, rtol=1e-6, atol=1e-8)
    r = res["lmi_lazy_stable"]
    assert r["has_x"] and r["niter"] < 400 and abs(r["niter"] - n_s) <= 2          # :213-217
    assert np.allclose(r["x"], xb_s, rtol=1e-6, atol=1e-8)
