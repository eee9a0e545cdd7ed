"""CPU: the oracle's LDLTMgr / LMIOracle / LMI0Oracle restatement (oracle/lmi_oracle.c) against every known answer
the reference holds for them: src/oracles/ldlt_mgr.rs:143-268, tests/lmi_tests.rs."""
import math

import numpy as np
import pytest

from oracle import oracle as O

CHOL1 = np.array([[25.0, 15.0, -5.0], [15.0, 18.0, 0.0], [-5.0, 0.0, 11.0]])
CHOL2 = np.array([[18.0, 22.0, 54.0, 42.0], [22.0, -70.0, 86.0, 62.0], [54.0, 86.0, -174.0, 134.0],
                  [42.0, 62.0, 134.0, -106.0]])
CHOL3 = np.array([[0.0, 15.0, -5.0], [15.0, 18.0, 0.0], [-5.0, 0.0, 11.0]])
CHOL7 = np.array([[0.0, 15.0, -5.0], [15.0, 18.0, 0.0], [-5.0, 0.0, -20.0]])
CHOL8 = np.array([[0.0, 15.0, -5.0], [15.0, 18.0, 0.0], [-5.0, 0.0, 20.0]])

F1 = np.array([[[-7.0, -11.0], [-11.0, 3.0]], [[7.0, -18.0], [-18.0, 8.0]], [[-2.0, -8.0], [-8.0, 1.0]]])
B1 = np.array([[33.0, -9.0], [-9.0, 26.0]])
F2 = np.array([[[-21.0, -11.0, 0.0], [-11.0, 10.0, 8.0], [0.0, 8.0, 5.0]],
               [[0.0, 10.0, 16.0], [10.0, -10.0, -10.0], [16.0, -10.0, 3.0]],
               [[-5.0, 2.0, -17.0], [2.0, -6.0, 8.0], [-17.0, 8.0, 6.0]]])
B2 = np.array([[14.0, 9.0, 40.0], [9.0, 91.0, 10.0], [40.0, 10.0, 15.0]])


def test_chol1_and_4():  # ldlt_mgr.rs:185-190, 211-216
    assert O.OracleLDLT(3).factorize(CHOL1)


def test_chol2_and_5():  # :192-199, 218-225
    m = O.OracleLDLT(4)
    assert not m.factorize(CHOL2)
    m.witness()
    assert m.pos == (0, 2)


def test_chol3():  # :201-209
    m = O.OracleLDLT(3)
    assert not m.factorize(CHOL3)
    ep = m.witness()
    assert m.pos == (0, 1) and m.wit[0] == 1.0 and ep == 0.0


def test_chol6_to_9():  # :227-255
    assert O.OracleLDLT(3).factor(lambda i, j: CHOL3[i, j], allow_semidefinite=True)
    m = O.OracleLDLT(3)
    assert not m.factor(lambda i, j: CHOL7[i, j], allow_semidefinite=True)
    assert m.witness() == pytest.approx(20.0, rel=1e-12)
    assert not O.OracleLDLT(3).factorize(CHOL8)
    assert O.OracleLDLT(3).factor(lambda i, j: CHOL8[i, j], allow_semidefinite=True)


def test_sqrt():  # :257-268
    mat = np.array([[1.0, 0.5, 0.5], [0.5, 1.25, 0.75], [0.5, 0.75, 1.5]])
    m = O.OracleLDLT(3)
    assert m.factor(lambda i, j: mat[i, j]) and m.is_spd()
    assert np.allclose(m.sqrt(), [[1.0, 0.5, 0.5], [0.0, 1.0, 0.5], [0.0, 0.0, 1.0]], rtol=0, atol=1e-15)


def test_factor_against_numpy_on_random_matrices():
    rng = np.random.default_rng(4)
    for nd in (1, 2, 5, 17, 40):
        a = rng.standard_normal((nd, nd))
        spd = a @ a.T + nd * np.eye(nd)
        m = O.OracleLDLT(nd)
        assert m.factorize(spd)
        L = np.tril(m.storage, -1) + np.eye(nd)
        D = np.diag(np.diag(m.storage))
        assert np.allclose(L @ D @ L.T, spd, rtol=1e-12, atol=1e-12)
        r = m.sqrt()
        assert np.allclose(r.T @ r, spd, rtol=1e-12, atol=1e-12)
        # make it indefinite: the witness certifies v'Av = -ep < 0
        bad = spd.copy()
        k = nd // 2
        bad[k, k] -= 2.0 * np.linalg.eigvalsh(spd)[-1] + 1.0
        assert not m.factorize(bad)
        ep = m.witness()
        s, e = m.pos
        v = np.zeros(nd)
        v[s:e] = m.wit[s:e]
        assert ep > 0 and v @ bad @ v == pytest.approx(-ep, rel=1e-9)
        assert m.sym_quad(bad) == pytest.approx(-ep, rel=1e-9)


def test_lmi_oracle_reference_points():  # tests/lmi_tests.rs:58-71
    assert O.OracleLMI(F1, B1).assess_feas(np.zeros(3)) is None
    assert O.OracleLMI(F1).assess_feas(np.zeros(3)) is not None


LMI0_F = np.array([[[1.0, 0.0], [0.0, 0.0]], [[0.0, 1.0], [1.0, 0.0]], [[0.0, 0.0], [0.0, 1.0]]])


def test_lmi0_reference_points():  # tests/lmi_tests.rs:77-113
    assert O.OracleLMI(LMI0_F).assess_feas(np.array([1.0, 0.0, 1.0])) is None
    cut = O.OracleLMI(LMI0_F).assess_feas(np.array([-1.0, 0.0, -1.0]))
    assert cut is not None
    g, ep = cut
    assert g[0] == pytest.approx(-1.0) and abs(g[1]) < 1e-12 and abs(g[2]) < 1e-12 and ep == pytest.approx(1.0)
    assert O.OracleLMI(LMI0_F).assess_feas(np.array([1.0, 1.0, 1.0])) is not None


class MyLmiOracle:
    """tests/lmi_tests.rs:121-171"""

    def __init__(self, make):
        self.idx = -1
        self.c = np.array([1.0, -1.0, 1.0])
        self.lmi1 = make(F1, B1)
        self.lmi2 = make(F2, B2)

    def assess_optim(self, xc, gamma):
        f0 = 0.0
        for a, b in zip(self.c.tolist(), np.asarray(xc).tolist()):
            f0 += a * b
        for _ in range(3):
            self.idx = 0 if self.idx == 2 else self.idx + 1
            if self.idx == 0:
                cut = self.lmi1.assess_feas(xc)
                if cut is not None:
                    return (cut[0], cut[1]), False, gamma
            elif self.idx == 1:
                cut = self.lmi2.assess_feas(xc)
                if cut is not None:
                    return (cut[0], cut[1]), False, gamma
            else:
                fj = f0 - gamma
                if fj > 0.0:
                    return (self.c.copy(), fj), False, gamma
                gamma = f0
        return (self.c.copy(), 0.0), True, gamma


def run_lmi(space, omega, max_iters=2000, tol=1e-20):
    """cutting_plane_optim with Options::default() (src/cutting_plane.rs:50-100, 286-313)"""
    gamma, x_best = math.inf, None
    for niter in range(max_iters):
        x = np.array(space.xc)
        (g, beta), shrunk, gamma = omega.assess_optim(x, gamma)
        if shrunk:
            x_best = x
            st = space.update_central_cut(g, beta)
        else:
            st = space.update_bias_cut(g, beta)
        if st != 0 or space.tsq < tol:
            return x_best, niter
    return x_best, max_iters


def test_lmi_lazy_runs():  # tests/lmi_tests.rs:199-217
    x_best, niter = run_lmi(O.OracleEll.new_with_scalar(10.0, np.zeros(3)), MyLmiOracle(O.OracleLMI))
    assert x_best is not None and niter < 300
    x_best, niter = run_lmi(O.OracleEllStable.new_with_scalar(10.0, np.zeros(3)), MyLmiOracle(O.OracleLMI))
    assert x_best is not None and niter < 400
