"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the reference's
python_ai `Ell._update_core` dense arithmetic): replayed through the oracle on CPU and through the HIP
engine on the GPU."""
import glob
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "ell_dense_*.npz")))
assert FILES, "golden fixtures missing"


def _replay(space, d, update, get_state, tol):
    for i in range(len(d["kinds"])):
        b1 = None if np.isnan(d["beta1"][i]) else float(d["beta1"][i])
        st = update(space, int(d["kinds"][i]), d["grads"][i], float(d["beta0"][i]), b1)
        assert int(st) == int(d["status"][i]), f"step {i}"
        xc, kappa, tsq = get_state(space)
        assert abs(tsq - d["tsq"][i]) <= tol * abs(d["tsq"][i]), f"tsq step {i}"
        assert abs(kappa - d["kappa"][i]) <= tol * abs(d["kappa"][i]), f"kappa step {i}"
        assert np.max(np.abs(xc - d["xc"][i])) <= tol * np.max(np.abs(d["xc"][i])), f"xc step {i}"


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_oracle_matches_golden(orc, path):
    d = np.load(path)
    n = int(d["n"])
    o = orc.OracleEll.new_with_matrix(float(d["kappa0"]), np.eye(n), d["xc0"])
    o.set_no_defer_trick(bool(d["no_defer"]))
    # numpy's BLAS sums in a different order than the reference's left fold: 1e-12, not bit-exact
    _replay(o, d, lambda s, k, g, b0, b1: s.update(k, g, b0, b1), lambda s: (np.array(s.xc), s.kappa, s.tsq), 1e-12)
    assert np.max(np.abs(o.mq - d["mq_final"])) <= 1e-12 * np.max(np.abs(d["mq_final"]))


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_hip_matches_golden(gpu, path):
    d = np.load(path)
    n = int(d["n"])
    e = gpu.Ell.new_with_matrix(float(d["kappa0"]), np.eye(n), d["xc0"])
    e.no_defer_trick = bool(d["no_defer"])
    _replay(e, d, lambda s, k, g, b0, b1: s._update(k, (g, (b0, b1))), lambda s: (s.xc(), s.kappa, s.tsq()), 1e-10)
    assert np.max(np.abs(e.mq - d["mq_final"])) <= 1e-10 * np.max(np.abs(d["mq_final"]))
