"""GPU: randomized walks over the C ABI's state machine, checked against the CPU oracle.

Every way of driving one search space -- synchronous updates, prime / cut / commit, the device-resident queue
(two-pass and pipelined, in pieces), failing cuts, clones, depth and no_defer_trick switches, set_xc, flush and the
observers of Q in the middle of all of it -- must describe the same ellipsoid as the reference's plain sequence of
`update_*_cut` calls (oracle/ell_oracle.c).  Seeds are fixed: a failure reproduces.
"""
import numpy as np
import pytest

from util import TOL, assert_state_close, beta_of, mixed_cut, random_factor, stable_tau, set_default

pytestmark = pytest.mark.gpu

# more walks for a soak run:  ELLHIP_FUZZ_SEEDS=200 python -m pytest tests/test_gpu_state_machine.py
import os
_EXTRA = int(os.environ.get("ELLHIP_FUZZ_SEEDS", "0"))


def _tau(o, g):
    if type(o).__name__ == "OracleEllStable":   # its packed buffer is not the shape matrix: ask a clone
        return stable_tau(o, g)
    return float(np.sqrt(max(o.kappa * (g @ (o.mq @ g)), 0.0)))


def _grad(rng, n):
    g = rng.standard_normal(n)
    return g / np.linalg.norm(g)


def _cut_for(o, rng, i, n, allow_fail=True):
    g = _grad(rng, n)
    kind, b0, b1 = mixed_cut(i if allow_fail else (i % 7), g, _tau(o, g), rng)   # i % 8 == 7 is the failing cut
    return kind, g, b0, b1


class Walk:
    def __init__(self, gpu, orc, n, seed, depths, stable=False):
        self.gpu, self.orc, self.n = gpu, orc, n
        self.rng = np.random.default_rng(seed)
        self.depths = depths
        self.stable = stable
        xc0 = self.rng.standard_normal(n)
        if stable:   # a NON-trivial factor: from the identity U and the scratch triangle stay exactly zero
            f = random_factor(n, 1000 + seed)
            self.g = gpu.EllStable.new_with_matrix(3.0, f, xc0)
            self.o = orc.OracleEllStable.new_with_matrix(3.0, f, xc0)
        else:
            self.g = gpu.Ell.new_with_scalar(3.0, xc0)
            self.o = orc.OracleEll.new_with_scalar(3.0, xc0)
        self.i = 0
        self.log = []

    # ---- observers (may run at any point of a pipelined / queued sequence)
    def observe(self, where):
        r = self.rng.integers(0, 7)
        self.log.append(f"observe{r}@{where}")
        if self.stable and r in (0, 4):
            r = 1
        if r == 0:
            self.g.flush()
        elif r == 1:
            q = self.g.mq
            assert np.max(np.abs(q - self.qref())) <= TOL * np.max(np.abs(self.qref())), self.log[-12:]
        elif r == 2:
            c = self.g.clone()
            assert self.stable or c.defer_depth == self.g.defer_depth
            del c
        elif r == 3:
            assert np.max(np.abs(self.g.xc() - self.o.xc)) <= TOL * max(np.max(np.abs(self.o.xc)), 1e-300), self.log[-12:]
        elif r == 4:   # (also while no_defer_trick is on: the depth is kept and takes effect when the flag goes off)
            self.g.defer_depth = int(self.rng.choice(self.depths))
        # 5, 6: nothing

    def qref(self):
        return self.o_mq_override if getattr(self, "o_mq_override", None) is not None else self.o.mq

    # ---- operations
    def op_update(self):
        m = int(self.rng.integers(1, 6))
        for _ in range(m):
            kind, g, b0, b1 = _cut_for(self.o, self.rng, self.i, self.n)
            self.i += 1
            so = self.o.update(kind, g, b0, b1)
            sg = self.g._update(kind, (g, beta_of(b0, b1)))
            assert int(sg) == so, self.log[-12:]
            assert abs(self.g.tsq() - self.o.tsq) <= TOL * abs(self.o.tsq), self.log[-12:]

    def op_pipelined(self):
        m = int(self.rng.integers(2, 12))
        # the gradients of a pipelined run are fixed before it starts (each is uploaded one cut ahead); the betas
        # are chosen when the cut is taken, from the oracle's state, as a caller would
        grads = [_grad(self.rng, self.n) for _ in range(m)]
        self.g.prime(grads[0])
        for j in range(m):
            if self.rng.random() < 0.3:
                self.observe("primed")
            kind, b0, b1 = mixed_cut(self.i, grads[j], _tau(self.o, grads[j]), self.rng)
            self.i += 1
            so = self.o.update(kind, grads[j], b0, b1)
            sg = self.g.cut(kind, (b0, b1))
            assert int(sg) == so, self.log[-12:]
            assert abs(self.g.tsq() - self.o.tsq) <= TOL * abs(self.o.tsq), self.log[-12:]
            if self.rng.random() < 0.15:   # observers between cut and commit see the shrunk Q
                q = self.g.mq
                assert np.max(np.abs(q - self.o.mq)) <= TOL * np.max(np.abs(self.o.mq)), self.log[-12:]
            self.g.commit(grads[j + 1] if j + 1 < m else None)

    def op_queue(self):
        m = int(self.rng.integers(3, 20))
        fail_at = int(self.rng.integers(0, m)) if self.rng.random() < 0.25 else -1
        kinds = np.zeros(m, dtype=np.int32)
        grads = np.empty((m, self.n))
        b0 = np.empty(m)
        b1 = np.full(m, np.nan)
        # the queue is fixed up front: betas are scaled by the CURRENT size of the ellipsoid (it shrinks slowly)
        tau0 = None
        for j in range(m):
            g = _grad(self.rng, self.n)
            if tau0 is None:
                tau0 = _tau(self.o, g)
            kind, c0, c1 = mixed_cut(self.i % 7, g, 0.5 * tau0, self.rng)
            self.i += 1
            if j == fail_at:
                kind, c0, c1 = 0, 50.0 * tau0, None
            kinds[j], grads[j], b0[j] = kind, g, c0
            if c1 is not None:
                b1[j] = c1
        self.g.queue_upload(kinds, grads, b0, b1)
        pos = 0
        while pos < m:
            step = int(self.rng.integers(1, m - pos + 1))
            self.g.queue_run(pos, step, fused=bool(self.rng.integers(0, 2)))
            pos += step
            if pos < m and self.rng.random() < 0.4:
                # the oracle has not caught up yet: observers that compare are given the state at `pos`
                self.catch_up(kinds, grads, b0, b1, pos)
                self.observe("queue")
        self.catch_up(kinds, grads, b0, b1, m)
        st, ts = self.g.queue_results()
        want = self.q_status
        assert list(st[:len(want)]) == want, (list(st), want, self.log[-12:])
        assert all(s == 3 for s in st[len(want):]), (list(st), want)   # cuts behind the halt never ran
        for j, t in enumerate(self.q_tsq):
            assert abs(ts[j] - t) <= TOL * abs(t), (j, self.log[-12:])
        self.q_done = 0

    def catch_up(self, kinds, grads, b0, b1, upto):
        """Advance the oracle over queue cuts [q_done, upto), stopping at the first failure like the device queue."""
        if not getattr(self, "q_done", 0):
            self.q_done, self.q_status, self.q_tsq, self.q_halted = 0, [], [], False
        while self.q_done < upto and not self.q_halted:
            j = self.q_done
            so = self.o.update(int(kinds[j]), grads[j], b0[j], None if np.isnan(b1[j]) else b1[j])
            self.q_status.append(so)
            self.q_tsq.append(self.o.tsq)
            self.q_halted = so != 0
            self.q_done += 1
        if self.q_halted:
            self.q_done = upto

    def op_set_xc(self):
        x = self.rng.standard_normal(self.n)
        self.g.set_xc(x)
        self.o.set_xc(x)

    def op_clone(self):
        self.g = self.g.clone()
        self.o = self.o.clone()

    def op_depth(self):
        self.g.defer_depth = int(self.rng.choice(self.depths))

    def op_no_defer_trick(self):
        # whatever depth is in force: the scaled data flow rewrites Q at every cut, so the recorded schedule is
        # suspended while the flag is on (ellhip.h) and resumes afterwards
        flag = not self.g.no_defer_trick
        self.g.no_defer_trick = flag
        self.o.set_no_defer_trick(flag)

    def run(self, nops):
        ops = [self.op_update, self.op_pipelined, self.op_queue, self.op_set_xc, self.op_clone, self.op_depth,
               self.op_no_defer_trick]
        weights = np.array([3, 4, 4, 1, 1, 0, 0] if self.stable else [3, 4, 4, 1, 1, 2, 1], dtype=float)
        for k in range(nops):
            op = ops[int(self.rng.choice(len(ops), p=weights / weights.sum()))]
            self.log.append(op.__name__ if self.stable else f"{op.__name__}[depth {self.g.defer_depth}]")
            self.q_done = 0
            op()
            if k % 3 == 2:
                assert_state_close(self.g, self.o, what=f"after op {k}: {self.log[-6:]}")
        assert_state_close(self.g, self.o, what=f"final: {self.log[-6:]}")


@pytest.mark.parametrize("seed", range(max(12, _EXTRA)))
@pytest.mark.parametrize("n", [40, 129, 640, 1024])
def test_random_walks_match_oracle(gpu, orc, n, seed, monkeypatch):
    set_default("SYMV_MIN_N", 512)   # n = 640, 1024 run the lower-triangle schedule at depth 8 / 16
    depths = (1, 8, 16, 24) if (n >= 512 and n % 2 == 0) else (1, 8)
    Walk(gpu, orc, n, 7000 + 31 * seed + n, depths).run(36)


@pytest.mark.parametrize("seed", range(max(6, _EXTRA // 2)))
@pytest.mark.parametrize("n", [40, 129, 300])
def test_random_walks_match_oracle_ellstable(gpu, orc, n, seed):
    """EllStable (src/ell_stable.rs) under the same drivers: the whole buffer -- diagonal, factor and the scratch
    triangle a failed cut rewrites -- stays equal to the oracle's."""
    Walk(gpu, orc, n, 9000 + 17 * seed + n, (1,), stable=True).run(24)


class ShardWalk:
    """The same idea for a row shard driven through ellalgo_rs_amd.sharded.ShardedEll (one rank, in process: the
    exchange is the identity, every code path of the shard -- two-phase updates, queue in pieces, dropped primes --
    is the multi-GPU one)."""

    def __init__(self, gpu, orc, n, seed, symmetric):
        from ellalgo_rs_amd.sharded import ShardedEll
        self.n, self.symmetric = n, symmetric
        self.rng = np.random.default_rng(seed)
        xc0 = self.rng.standard_normal(n)
        self.g = ShardedEll.new_with_scalar(3.0, xc0, rank=0, world=1, exchange=lambda gt, row0, nrows: None,
                                            symmetric=symmetric)
        self.o = orc.OracleEll.new_with_scalar(3.0, xc0)
        self.depths = (8, 16, 24) if symmetric else (1, 8)
        self.i = 0
        self.log = []

    def check(self, what):
        q, qo = self.g.mq_rows, self.o.mq
        if self.symmetric:   # rows are current up to their diagonal only
            q, qo = np.tril(q), np.tril(qo)
        assert np.max(np.abs(q - qo)) <= TOL * np.max(np.abs(qo)), (what, self.log[-10:])
        assert np.max(np.abs(self.g.xc() - self.o.xc)) <= TOL * np.max(np.abs(self.o.xc)), (what, self.log[-10:])
        assert abs(self.g.kappa - self.o.kappa) <= TOL * abs(self.o.kappa), (what, self.log[-10:])

    def observe(self):
        r = int(self.rng.integers(0, 5))
        self.log.append(f"observe{r}")
        if r == 0:
            self.g.flush()
        elif r == 1:
            self.check("observer")
        elif r == 2:
            self.g.set_defer_depth(int(self.rng.choice(self.depths)))

    def op_update(self):
        for _ in range(int(self.rng.integers(1, 6))):
            g = _grad(self.rng, self.n)
            kind, b0, b1 = mixed_cut(self.i, g, _tau(self.o, g), self.rng)
            self.i += 1
            so = self.o.update(kind, g, b0, b1)
            assert int(self.g._update(kind, (g, beta_of(b0, b1)))) == so, self.log[-10:]
            assert abs(self.g.tsq() - self.o.tsq) <= TOL * abs(self.o.tsq), self.log[-10:]

    def op_queue(self):
        m = int(self.rng.integers(3, 20))
        fail_at = int(self.rng.integers(0, m)) if self.rng.random() < 0.25 else -1
        kinds = np.zeros(m, dtype=np.int32)
        grads = np.empty((m, self.n))
        b0 = np.empty(m)
        b1 = np.full(m, np.nan)
        tau0 = None
        for j in range(m):
            g = _grad(self.rng, self.n)
            if tau0 is None:
                tau0 = _tau(self.o, g)
            kind, c0, c1 = mixed_cut(self.i % 7, g, 0.5 * tau0, self.rng)
            self.i += 1
            if j == fail_at:
                kind, c0, c1 = 0, 50.0 * tau0, None
            kinds[j], grads[j], b0[j] = kind, g, c0
            if c1 is not None:
                b1[j] = c1
        self.g.queue_upload(kinds, grads, b0, b1)
        want, halted, done = [], False, 0

        def catch_up(upto):
            nonlocal halted, done
            while done < upto and not halted:
                so = self.o.update(int(kinds[done]), grads[done], b0[done], None if np.isnan(b1[done]) else b1[done])
                want.append(so)
                halted = so != 0
                done += 1

        pos = 0
        while pos < m:
            step = int(self.rng.integers(1, m - pos + 1))
            self.g.queue_run(pos, step, fused=bool(self.rng.integers(0, 2)))
            pos += step
            if pos < m and self.rng.random() < 0.4:
                catch_up(pos)
                self.observe()
        catch_up(m)
        st, _ = self.g.queue_results()
        assert list(st[:len(want)]) == want and all(s == 3 for s in st[len(want):]), (list(st), want, self.log[-10:])

    def op_set_xc(self):
        x = self.rng.standard_normal(self.n)
        self.g.set_xc(x)
        self.o.set_xc(x)

    def op_depth(self):
        self.g.set_defer_depth(int(self.rng.choice(self.depths)))

    def run(self, nops):
        ops = [self.op_update, self.op_queue, self.op_set_xc, self.op_depth]
        weights = np.array([3.0, 4.0, 1.0, 2.0])
        for k in range(nops):
            op = ops[int(self.rng.choice(len(ops), p=weights / weights.sum()))]
            self.log.append(op.__name__)
            op()
            if k % 3 == 2:
                self.check(f"after op {k}")
        self.check("final")


@pytest.mark.parametrize("seed", range(max(5, _EXTRA // 2)))
@pytest.mark.parametrize("n,symmetric", [(96, False), (640, False), (640, True), (1024, True)])
def test_random_walks_match_oracle_row_shard(gpu, orc, n, symmetric, seed):
    ShardWalk(gpu, orc, n, 11000 + 13 * seed + n + int(symmetric), symmetric).run(24)
