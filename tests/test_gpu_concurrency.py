"""GPU: several handles driven from several host threads at once on one card -- what BSearchAdaptor's clone per probe
(src/cutting_plane.rs:409-418) turns into when probes run in parallel.  Every kernel family that waits inside a launch is in the
mix: resident batches (a grid barrier per cut: cooperative launch, one grid at a time, commit-or-abandon), EllStable's persistent
solves (workgroups wait for workgroups dispatched before them; two handles of them), and the streamed queue runs whose matrix-core
passes keep every CU for a whole pass (k_symm_mfma_q).  Grids that wait inside a launch are chained per device (CoresScope).  Whatever the
interleaving, every handle must end in the state the oracle's plain sequence of updates gives (1e-10); a resident batch may be
abandoned and rerun (counted), nothing may fail."""
import threading

import numpy as np
import pytest

from test_gpu_resident import _cuts
from util import TOL, assert_state_close, oracle_update, random_factor

pytestmark = pytest.mark.gpu


def test_five_threads_four_kernel_families(gpu, orc):
    from ellalgo_rs_amd import synth
    jobs = []

    def resident_job(n, k, piece, seed):
        kinds, grads, b0, b1 = _cuts(n, k, seed)
        e = gpu.Ell.new_with_scalar(1.0, np.linspace(-1.0, 1.0, n))
        e.queue_upload(kinds, grads, b0, b1)

        def run():
            for a in range(0, k, piece):
                e.queue_run(a, piece, fused=True)

        def check():
            o = orc.OracleEll.new_with_scalar(1.0, np.linspace(-1.0, 1.0, n))
            st, ts = e.queue_results()
            assert np.all(st == 0)
            for i in range(k):
                assert oracle_update(o, int(kinds[i]), grads[i], b0[i], None if np.isnan(b1[i]) else b1[i]) == 0
                assert abs(ts[i] - o.tsq) <= TOL * abs(o.tsq), (n, i)
            assert_state_close(e, o, what=f"resident n={n} (abandoned {e.get_option(gpu.capi.OPT_RESIDENT_ABANDONED)})")
        return run, check

    def stable_job(n, k, seed):
        kinds, grads, b0, _ = synth.deep_cuts(n, k)
        f = random_factor(n, seed)
        e = gpu.EllStable.new_with_matrix(1.0, f, np.zeros(n))
        stats = []

        def run():
            for i in range(k):
                stats.append(int(e.update_bias_cut((grads[i], float(b0[i])))))

        def check():
            o = orc.OracleEllStable.new_with_matrix(1.0, f, np.zeros(n))
            for i in range(k):
                assert o.update(0, grads[i], b0[i]) == stats[i] == 0
            assert_state_close(e, o, what=f"EllStable n={n}")
        return run, check

    def queue_job(n, k, seed):
        kinds, grads, b0, b1 = synth.parallel_cuts(n, k)
        e = gpu.Ell.new_with_scalar(1.0, np.zeros(n))
        e.queue_upload(kinds, grads, b0, b1)

        def run():
            for a in range(0, k, 20):
                e.queue_run(a, min(20, k - a), fused=True)

        def check():
            o = orc.OracleEll.new_with_scalar(1.0, np.zeros(n))
            st, ts = e.queue_results()
            assert np.all(st == 0)
            for i in range(k):
                assert o.update_rowwise_mt(int(kinds[i]), grads[i], b0[i], b1[i]) == 0
                assert abs(ts[i] - o.tsq) <= TOL * abs(o.tsq), i
            assert_state_close(e, o, what=f"queue run n={n}")
        return run, check

    jobs.append(resident_job(4096, 72, 12, 11))
    jobs.append(resident_job(2048, 96, 8, 12))
    jobs.append(stable_job(4096, 40, 13))
    jobs.append(stable_job(3072, 50, 15))   # (two EllStable handles: two sets of persistent solves -- chained per device, CoresScope)
    jobs.append(queue_job(8192, 60, 14))
    start = threading.Barrier(len(jobs))
    errs = []

    def worker(run):
        try:
            start.wait()
            run()
        except Exception as ex:   # noqa: BLE001 -- reported below
            errs.append(repr(ex))

    ts = [threading.Thread(target=worker, args=(run,)) for run, _ in jobs]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for _, check in jobs:
        check()
