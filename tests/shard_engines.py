"""TEST DOUBLE for the per-rank engine of ellalgo_rs_amd.sharded.ShardedEll: the row-block pieces of
the update computed with the CPU oracle / exact numpy elementwise arithmetic, so the multi-rank
orchestration (partition, in-place all-gather, redundant scalar stage, local rank-1) can be checked
for bit-equality against the single-process oracle with the gloo backend and no GPU."""
import numpy as np
import torch

from oracle import oracle


class OracleShardEngine:
    def __init__(self, n, row0, nrows, kappa, mq_rows, diag, xc):
        self.n, self.row0, self.nrows = n, row0, nrows
        if mq_rows is not None:
            self.Q = np.array(mq_rows, dtype=np.float64).reshape(nrows, n)
        else:
            self.Q = np.zeros((nrows, n))
            for r in range(nrows):
                self.Q[r, row0 + r] = 1.0 if diag is None else diag[row0 + r]
        self.xcv = np.array(xc, dtype=np.float64)
        self.kap, self.tsqv = float(kappa), 0.0
        self.gt = torch.zeros(n, dtype=torch.float64)
        self.gt_np = self.gt.numpy()  # shares memory with the tensor the collective fills
        self.calc = oracle.Calc(n)
        self.cut = None
        self.q = None
        self.halted = False
        self.pending = None
        self.primed = -1            # queue index whose GEMV is in place (ellhip_queue_primed)
        self.drop_on_flush = False  # emulate a shard on a recorded schedule: flush / mq_rows / a depth switch apply
                                    # recorded updates, which invalidates a primed GEMV (the real library drops it)

    symmetric = False

    def set_symmetric(self, flag=True):
        """ellhip_set_shard_symmetric: the GEMV yields this shard's partial sums over its lower trapezoid (the
        double keeps its full rows current; only the GEMV's data flow is what the orchestration test is about)."""
        self.symmetric = bool(flag)

    def set_defer_depth(self, depth):
        self._observer()

    # ---- the three primitives (mirror of prime / cut / commit in csrc/ellhip_capi.hip)
    def _gemv(self, g):
        if not self.symmetric:
            oracle.rows_gemv(self.n, self.row0, self.nrows, self.Q, g, self.gt_np)
            return
        y = np.zeros(self.n)
        for r in range(self.nrows):
            i = self.row0 + r
            row = self.Q[r, :i + 1]
            y[i] += float(row @ g[:i + 1])          # row sums, columns up to the diagonal
            y[:i] += row[:i] * g[i]                  # column sums, strictly below the diagonal
        self.gt_np[:] = y                            # partial: the all-reduce adds the shards' vectors

    def _scalar(self) -> int:
        kind, g, b0, b1 = self.cut
        gt = self.gt_np
        omega = 0.0
        for a, b in zip(g, gt):          # Arr::dot: left fold (src/arr.rs:443-451)
            omega += float(a) * float(b)
        self.tsqv = self.kap * omega
        st, (rho, sigma, delta) = self.calc.dispatch(kind, b0, b1, self.tsqv)
        self.pending = None
        if st != 0:
            return st
        self.xcv -= (rho / omega) * gt   # every rank updates its full copy identically
        self.pending = (sigma / omega, gt.copy())
        self.kap *= delta
        return st

    def _shrink(self):
        if self.pending is None:
            return
        ratio, gt = self.pending
        self.pending = None
        rows = self.row0 + np.arange(self.nrows)[:, None]
        cols = np.arange(self.n)[None, :]
        lower = (ratio * gt[rows]) * gt[cols]      # (ratio*gt[i])*gt[j], j <= i   (src/ell.rs:119-121)
        upper = (ratio * gt[cols]) * gt[rows]      # mirrored element (i, j) = (col, row)
        self.Q -= np.where(cols <= rows, lower, upper)

    # ---- phase 1 / phase 2 (mirror of ellhip_update_begin / ellhip_update_end)
    def begin(self, kind, g, b0, has1, b1):
        self.cut = (kind, np.array(g, dtype=np.float64), b0, b1 if has1 else None)
        self._gemv(self.cut[1])

    def end(self) -> int:
        st = self._scalar()
        self._shrink()
        return st

    # ---- queue
    def queue_upload(self, k, kinds, grads, b0, has1, b1):
        self.q = (kinds.copy(), grads.reshape(k, self.n).copy(), b0.copy(), has1.copy(), b1.copy())
        self.qstatus = np.full(k, -1, dtype=np.int32)
        self.qtsq = np.zeros(k)
        self.halted = False

    def _qcut(self, i):
        kinds, grads, b0, has1, b1 = self.q
        return (int(kinds[i]), grads[i], float(b0[i]), float(b1[i]) if has1[i] else None)

    def queue_prime(self, i):
        if not self.halted:
            self.cut = self._qcut(i)
            self._gemv(self.cut[1])
            self.primed = i

    def queue_cut(self, i):
        if self.halted:
            self.qstatus[i] = 3
            return
        assert self.primed == i, f"cut {i} taken without its GEMV in place (primed: {self.primed})"
        self.primed = -1
        self.cut = self._qcut(i)
        st = self._scalar()
        self.qstatus[i], self.qtsq[i] = st, self.tsqv
        if st != 0:
            self.halted = True

    def queue_commit(self, i, nxt):
        if self.halted:
            return
        self._shrink()
        if nxt >= 0:
            self._gemv(self._qcut(nxt)[1])   # GEMV of the next cut on the freshly shrunk rows
            self.primed = nxt

    def queue_begin(self, i):
        self.queue_prime(i)

    def queue_end(self, i):
        self.queue_cut(i)
        self.queue_commit(i, -1)

    def queue_results(self, k):
        self.halted = False
        return self.qstatus.copy(), self.qtsq.copy()

    def xc(self): return self.xcv.copy()
    def set_xc(self, x): self.xcv = np.array(x, dtype=np.float64)
    def mq_rows(self):
        self._observer()
        return self.Q.copy()
    def kappa(self): return self.kap
    def tsq(self): return self.tsqv
    def synchronize(self): pass

    def queue_primed(self) -> int:
        return self.primed

    def _observer(self):
        if self.drop_on_flush and self.primed >= 0:
            self.primed = -1
            self.gt_np[:] = np.nan   # whoever uses this vector without priming (and exchanging) again is caught

    def flush(self):
        self._observer()
