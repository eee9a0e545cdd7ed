#!/usr/bin/env python3
"""bench.py -- ellipsoid updates/sec on MI355X (BASELINE.json metric), one JSON line on stdout.

A "step" is one ellipsoid update (Ell::update_core, src/ell.rs:97-137) = GEMV pass + scalar stage +
rank-1 pass over the n*n f64 matrix, driven from a device-resident queue of synthetic cuts
(SURVEY.md 8d), so all inputs are in HBM when the timed region starts.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

N > 1: Q is row-block partitioned over the ranks (same n: strong scaling); every update does one
in-place RCCL all-gather of the n-vector Q*g between the two passes.

Besides the contract fields the line carries
  roofline     : HBM roofline of the dominant kernel (k_rank1, 16*n^2 algorithmic bytes per launch),
                 its duration measured with HIP events on the launch stream; per-kernel and
                 whole-update (24*n^2 B) figures alongside.
  cpu_baseline : the CPU oracle (reference loop order, 1 thread) timed on this box, rank 0, N = 1.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_MFMA_PEAK_TF = 78.6  # v_mfma_f64_16x16x4_f64: 64 cycles per instruction and SIMD = the FP64 vector-FMA rate, 1024 SIMDs, 2.4 GHz
                          # (measured back to back on one accumulator: 68 cycles, 71.6-75.2 TFLOP/s: profiles/r04/mfma_f64_rate.txt)


def product_pass_roof(n2w: float, gradients_per_pass: float, avg_ms: float) -> dict:
    """Which roof binds the matrix-core product pass (k_symm_mfma_q / _q2), and how far below it a launch of `avg_ms` is.
    The pass multiplies every block of the lower triangle with the group's gradients twice (column and row product): 2 n^2 flop
    per gradient for 4 n^2 bytes read, i.e. (gradients per pass) / 2 flop per byte, which crosses this card's ridge
    (78.6 TFLOP/s / 8 TB/s = 9.8 flop/B) at 19.65 gradients per pass.  n2w: the matrix elements one GPU holds (n^2 / ranks).
    Padding columns of the 16-wide MFMA tiles are NOT counted as flop."""
    flop = 2.0 * n2w * gradients_per_pass
    tf = flop / (avg_ms * 1e-3) / 1e12
    gbps = 4.0 * n2w / (avg_ms * 1e-3) / 1e9
    intensity = gradients_per_pass / 2.0
    ridge = FP64_MFMA_PEAK_TF * 1e12 / (HBM_PEAK_GBS * 1e9)
    return {"bound": "mfma" if intensity > ridge else "hbm", "flop_per_launch": flop, "flop_per_byte": intensity, "ridge": ridge,
            "TFLOPs": tf, "mfma_frac": tf / FP64_MFMA_PEAK_TF, "GBps": gbps, "hbm_frac": gbps / HBM_PEAK_GBS}

WORKLOADS = {
    # name: (n, variant, cut generator, description)
    "n16384-parallel": (16384, "ell", "parallel", "config 3: n=16384 Ell, alternating parallel-central / parallel-bias cuts"),
    "n16384-deep": (16384, "ell", "deep", "n=16384 Ell, deep cuts beta~U[0,0.1)"),
    "n4096-deep": (4096, "ell", "deep", "config 2: n=4096 Ell, deep cuts (Q=128 MiB fits the 256 MiB Infinity Cache)"),
    "n8192-deep": (8192, "ell", "deep", "n=8192 Ell, deep cuts"),
    "n32768-deep": (32768, "ell", "deep", "config 4: n=32768 Ell, deep cuts (8 GiB Q)"),
    "n16384-ellstable": (16384, "ellstable", "deep", "config 5: n=16384 EllStable (random unit-triangular factor), deep cuts"),
    "n4096-ellstable": (4096, "ellstable", "deep", "n=4096 EllStable (random unit-triangular factor), deep cuts"),
}


# the reference's large-n caller of the path (LowpassOracle + cutting_plane_optim), run as
# a device-resident loop.  Not the headline metric: `--workload lowpass-n4096` prints its own JSON line.
LOWPASS_WORKLOADS = {"lowpass-n4096": 4096, "lowpass-n2048": 2048, "lowpass-n1024": 1024, "lowpass-n256": 256}


# BASELINE.json configs 2, 3, 4 (its one-GPU form: the whole 8 GiB matrix on one card) and 5
ALL_CONFIGS = ["n4096-deep", "n16384-parallel", "n32768-deep", "n16384-ellstable"]


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# many independent small ellipsoids (BASELINE config 1 is n = 16) in one launch.
BATCH_WORKLOADS = {"batch-n16": (16, 65536, 8), "batch-n32": (32, 32768, 8), "batch-n64": (64, 8192, 8),
                   "batch-n128": (128, 2048, 8)}


def batch_bench(args, real_stdout) -> None:
    """B ellipsoids of dimension n, K central cuts each per launch (include/ellhip_batch.h), all inputs resident
    in HBM; value = ellipsoid updates/s summed over the batch."""
    import torch
    if args.gpus != 1 or int(os.environ.get("WORLD_SIZE", "1")) != 1:
        raise SystemExit("independent ellipsoids: replicas only (run with --gpus 1)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU path")
    import ellalgo_rs_amd as pkg
    n, B, K = BATCH_WORKLOADS[args.workload]
    S, W = args.steps, args.warmup
    rng = np.random.default_rng(0x5EED)
    grads = rng.standard_normal((K, B, n))
    grads /= np.linalg.norm(grads, axis=2, keepdims=True)
    kinds = np.full((K, B), 1, dtype=np.int32)  # update_central_cut: succeeds however small the ellipsoid gets
    b0 = np.zeros((K, B))
    hb1 = np.zeros((K, B), dtype=np.int32)
    b1 = np.zeros((K, B))
    batch = pkg.EllBatch.new_with_scalar(1.0, np.zeros((B, n)))
    dev = torch.device("cuda", 0)
    t = {k: torch.from_numpy(v).to(dev) for k, v in dict(kinds=kinds, grads=grads, b0=b0, hb1=hb1, b1=b1).items()}
    status = torch.full((K, B), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def launch():
        batch.update_dev(K, t["kinds"].data_ptr(), t["grads"].data_ptr(), t["b0"].data_ptr(), t["hb1"].data_ptr(),
                         t["b1"].data_ptr(), status.data_ptr(), None)

    for _ in range(W):
        launch()
    batch.synchronize()
    t0 = time.perf_counter()
    for _ in range(S):
        launch()
    batch.synchronize()
    elapsed = time.perf_counter() - t0
    assert bool((status == 0).all().item()), "a cut did not succeed"
    # kernel duration with HIP events on the launch stream (torch.cuda.Event only sees torch's stream)
    stream = torch.cuda.ExternalStream(batch._lib.ellhip_batch_stream(batch._h))
    evs = []
    for _ in range(min(S, 20)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        launch()
        b.record(stream)
        evs.append((a, b))
    batch.synchronize()
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    alg = B * (16.0 * n * n + K * (n * 8.0 + 24.0) + 16.0 * n + 32.0)
    gbps = alg / (kern_ms * 1e-3) / 1e9
    out = {
        "metric": "ellipsoid updates/sec, %d independent ellipsoids of n=%d (batched engine)" % (B, n),
        "value": S * K * B / elapsed, "unit": "updates/s", "n_gpus": 1, "steps": S, "warmup": W,
        "ms_per_step": elapsed / S * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": args.workload, "n": n, "ellipsoids": B, "cuts_per_launch": K, "space": "ell",
                   "cuts": "central, random unit gradients", "q_bytes_total": 8.0 * n * n * B},
        "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel": "k_batch_update",
                     "achieved": gbps, "frac": gbps / HBM_PEAK_GBS, "traffic": None, "alg_bytes_per_launch": alg,
                     "avg_launch_ms": kern_ms,
                     "byte_model": "per ellipsoid and launch: Q in + out 16*n^2, K gradients K*n*8, cut values and "
                                   "results K*24, xc in + out 16*n: the matrix stays in LDS for the K cuts"},
    }
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f).get(args.workload)
        if pmc:
            out["roofline"]["traffic"] = pmc.get("batch")
            out["roofline"]["traffic_source"] = pmc.get("source")
    except OSError:
        pass
    if not args.no_cpu_baseline:
        from oracle import oracle as O
        log("[batch] timing the CPU oracle (bounded sample) ...")
        Bs = min(B, 2048)
        sl = (slice(None), slice(0, Bs))
        O.ell_batch_run(kinds[sl], grads[sl], b0[sl], hb1[sl], b1[sl], want_state=False)  # warm
        done, t_used = 0, 0.0
        while t_used < args.cpu_budget:
            t1 = time.perf_counter()
            ok, _, _, _, _ = O.ell_batch_run(kinds[sl], grads[sl], b0[sl], hb1[sl], b1[sl], want_state=False)
            t_used += time.perf_counter() - t1
            assert ok == K * Bs
            done += K * Bs
        out["cpu_baseline"] = {"value": done / t_used, "unit": "updates/s", "cores": 1, "kind": "port",
                               "sample": f"{done} updates: {Bs} of the ellipsoids x {K} cuts, repeated, oracle/ell_oracle.c "
                                         f"orc_ell_batch_run, 1 thread, {t_used:.1f} s", "host_cpus": os.cpu_count()}
    print(json.dumps(out), file=real_stdout, flush=True)


# LMIOracle / LDLTMgr on the device.
LMI_WORKLOADS = {"lmi-m2048-n64": (64, 2048), "lmi-m1024-n128": (128, 1024), "lmi-m4096-n16": (16, 4096)}


def lmi_bench(args, real_stdout) -> None:
    """LMIOracle::assess_feas calls/s for an m x m pencil with n variables (include/ellhip_lmi.h): a feasible point
    (full LDL^T: the worst case) and an infeasible one (factorisation stops at the failing pivot, cut produced).
    value = feasible-point calls/s.  The F_k stream is priced against HBM; the factorisation is O(m^3/3) of
    two-rounding multiply-adds in the reference's summation order (not an MFMA shape: every element owns an
    ordered accumulator)."""
    import torch
    if args.gpus != 1 or int(os.environ.get("WORLD_SIZE", "1")) != 1:
        raise SystemExit("one oracle per GPU: replicas only (run with --gpus 1)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU path")
    import ellalgo_rs_amd as pkg
    n, m = LMI_WORKLOADS[args.workload]
    S, W = args.steps, args.warmup
    rng = np.random.default_rng(0x5EED)
    a = rng.standard_normal((m, 64))
    B = a @ a.T / 64.0 + 1.0 * np.eye(m)
    F = rng.standard_normal((n, m, m))
    F = (F + F.transpose(0, 2, 1)) / 2.0
    x_feas = np.zeros(n)
    x_dir = rng.standard_normal(n) / math.sqrt(n * m)
    t0 = time.perf_counter()
    dev = pkg.LMIOracle(F, B)
    log(f"[lmi] {n} matrices {m} x {m} ({n * m * m * 8 / 2**30:.2f} GiB) uploaded in {time.perf_counter() - t0:.1f}s")
    assert dev.assess_feas(x_feas) is None
    scale = 0.25  # walk out along x_dir until F(x) stops being positive definite
    while dev.assess_feas(scale * x_dir) is None:
        scale *= 1.5
        assert scale < 1e6
    x_cut = scale * x_dir
    cut = dev.assess_feas(x_cut)
    p_cut = dev.pos[1]

    def timed(x, count):
        for _ in range(W):
            dev.assess_feas(x)
        t1 = time.perf_counter()
        for _ in range(count):
            dev.assess_feas(x)
        return (time.perf_counter() - t1) / count

    t_feas = timed(x_feas, S)
    t_cut = timed(x_cut, S)
    form_bytes = n * m * (m + 1) / 2 * 8.0
    flops = m ** 3 / 3.0 * 2.0
    out = {
        "metric": "LMIOracle assess_feas calls/sec, %d x %d pencil with %d variables (feasible point, full LDL^T)" % (m, m, n),
        "value": 1.0 / t_feas, "unit": "calls/s", "n_gpus": 1, "steps": S, "warmup": W, "ms_per_step": t_feas * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": args.workload, "n": n, "m": m, "pencil_bytes": 8.0 * n * m * m,
                   "cut_case": {"ms_per_call": t_cut * 1e3, "failing_row": p_cut,
                                "note": "factorisation stops at the failing pivot; witness + n quadratic forms follow"}},
        "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel": "whole call", "traffic": None,
                     "achieved": form_bytes / t_feas / 1e9, "frac": form_bytes / t_feas / 1e9 / HBM_PEAK_GBS,
                     "byte_model": "lower triangle of the n matrices read once to form F(x): n*m*(m+1)/2*8 B; the call is "
                                   "dominated by the ordered LDL^T, see factor_gflops",
                     "factor_gflops": flops / t_feas / 1e9,
                     "whole_update": {"frac": form_bytes / t_feas / 1e9 / HBM_PEAK_GBS}},
    }
    if not args.no_cpu_baseline:
        from oracle import oracle as O
        log("[lmi] timing the CPU oracle (bounded sample) ...")
        cpu = O.OracleLMI(F, B)
        t1 = time.perf_counter()
        assert cpu.assess_feas(x_feas) is None
        t_cpu = time.perf_counter() - t1
        t1 = time.perf_counter()
        rc = cpu.assess_feas(x_cut)
        t_cpu_cut = time.perf_counter() - t1
        assert rc is not None and cpu.ldlt.pos[1] == p_cut and rc[1] == cut[1].beta
        out["cpu_baseline"] = {"value": 1.0 / t_cpu, "unit": "calls/s", "cores": 1, "kind": "port",
                               "sample": f"1 feasible-point call ({t_cpu:.2f} s) and 1 cut call ({t_cpu_cut:.2f} s, same failing "
                                         f"row {p_cut} and bit-identical ep) of oracle/lmi_oracle.c, 1 thread",
                               "host_cpus": os.cpu_count(), "cut_case_calls_per_s": 1.0 / t_cpu_cut}
    print(json.dumps(out), file=real_stdout, flush=True)


def lowpass_bench(args, real_stdout) -> None:
    """cutting_plane_optim(LowpassOracle, Ell) iterations/s with everything on the device
    (include/ellhip_lowpass.h), the oracle's scan priced against the HBM roofline with the rows the
    reference's walk visits as algorithmic bytes, and the CPU oracle loop timed on the first iterations."""
    import torch
    if args.gpus != 1 or int(os.environ.get("WORLD_SIZE", "1")) != 1:
        raise SystemExit("the device-resident loop does not shard: replicas only (run it with --gpus 1)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU path")
    import ellalgo_rs_amd as pkg
    n = LOWPASS_WORKLOADS[args.workload]
    K, W, P = args.steps, args.warmup, args.profile_steps
    dep = args.defer if args.defer in (1, 8) else 8  # (n <= 4096 here: the full-row schedule, depth 8)
    c = pkg.lowpass_case_constants(corrected=True)
    kappa0 = 40.0

    def fresh():
        t0 = time.perf_counter()
        o = pkg.LowpassOracle(n, *c)
        sp = pkg.Ell.new_with_scalar(kappa0, np.zeros(n))
        sp.defer_depth = dep
        log(f"[lowpass] table {15 * n} x {n} ({15 * n * n * 8 / 2**30:.2f} GiB) + Q built in {time.perf_counter() - t0:.1f}s")
        return o, sp

    oracle, space = fresh()
    gamma = c[4]
    _, nit, gamma = oracle.cutting_plane_optim(space, gamma, W, 0.0)
    assert nit == W, f"warm-up stopped after {nit} iterations"
    oracle.rows_visited(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, nit, gamma = oracle.cutting_plane_optim(space, gamma, K, 0.0)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if nit != K:
        raise SystemExit(f"the run reached its natural end after {W + nit} iterations (no further improvement of gamma "
                         f"is possible: NoSoln): use fewer --steps for {args.workload}")
    rows_timed = oracle.rows_visited(reset=True)

    per_kernel, rows_prof = {}, 0
    if P > 0:
        space.profile_enable(True)
        _, nit, gamma = oracle.cutting_plane_optim(space, gamma, P, 0.0)
        assert nit == P
        prof = space.profile_read()
        space.profile_enable(False)
        rows_prof = oracle.rows_visited(reset=True)
        for name, (ms, cnt) in prof.items():
            if cnt:
                per_kernel[name] = {"avg_ms": ms / cnt, "launches": cnt}
    n2 = float(n) * n
    scan_bytes_prof = rows_prof / max(P, 1) * n * 8.0
    alg = {"lp_scan": scan_bytes_prof, "gemv": 8.0 * n2, "symv": 4.0 * n2, "rank1": 16.0 * n2, "fused": 16.0 * n2,
           "apply": 16.0 * n2, "apply_gemv": 16.0 * n2}
    for name, e in per_kernel.items():
        if name in alg:
            e["alg_bytes"] = alg[name]
            e["GBps"] = alg[name] / (e["avg_ms"] * 1e-3) / 1e9
    ms_per_step = elapsed / K * 1e3
    roofline = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None, "kernel": "k_lp_scan",
                "byte_model": "rows the reference's walk visits (early exit at the first violated constraint) x n x 8 B; "
                              "the device reads those rows plus at most one grid-wide round of 16-row chunks",
                "per_kernel": per_kernel}
    if "lp_scan" in per_kernel:
        roofline.update({"achieved": per_kernel["lp_scan"]["GBps"], "frac": per_kernel["lp_scan"]["GBps"] / HBM_PEAK_GBS,
                         "alg_bytes_per_launch": scan_bytes_prof, "avg_launch_ms": per_kernel["lp_scan"]["avg_ms"],
                         "rows_visited_per_call": rows_prof / max(P, 1), "rows_in_table": 15 * n})
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f).get(args.workload)
        if pmc:
            roofline["traffic"] = pmc.get("lp_scan")
            roofline["traffic_source"] = pmc.get("source")
    except OSError:
        pass
    upd_bytes = {1: 16.0, 8: 9.0}[dep] * n2
    whole = rows_timed / K * n * 8.0 + upd_bytes
    roofline["whole_iteration"] = {"alg_bytes": whole, "GBps": whole / (ms_per_step * 1e-3) / 1e9,
                                   "frac": whole / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "model": f"oracle rows {rows_timed / K:.0f} x n x 8 + ellipsoid update "
                                            f"{upd_bytes / n2:.0f}*n^2 (defer depth {dep}, shrink fused with the next GEMV)"}
    out = {
        "metric": "cutting-plane iterations/sec, LowpassOracle + Ell device-resident loop at n=%d" % n,
        "value": K / elapsed, "unit": "iterations/s", "n_gpus": 1, "steps": K, "warmup": W,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": args.workload, "n": n, "space": "ell", "defer_depth": dep,
                   "oracle": "LowpassOracle, corrected create_lowpass_case constants (parity unpinned by the reference, "
                             "see oracle/lowpass_oracle.h)", "table_bytes": 15.0 * n2 * 8.0, "kappa0": kappa0,
                   "gamma_after": gamma},
        "roofline": roofline,
    }
    if not args.no_cpu_baseline:
        from oracle import oracle as O
        log("[lowpass] timing the CPU oracle loop (bounded sample) ...")
        t0 = time.perf_counter()
        co = O.OracleLowpass(n, *c)
        ce = O.OracleEll.new_with_scalar(kappa0, np.zeros(n))
        t_init = time.perf_counter() - t0
        cg, done, t_used = c[4], 0, 0.0
        step = 4 if n >= 2048 else 64
        while t_used < args.cpu_budget:
            t1 = time.perf_counter()
            _, nit, cg, last = co.cutting_plane_optim(ce, cg, step, 0.0)
            t_used += time.perf_counter() - t1
            assert nit == step and last == 0
            done += step
        # the same first iterations on the device, from a fresh start
        o2, s2 = fresh()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        _, nit, g2 = o2.cutting_plane_optim(s2, c[4], done, 0.0)
        torch.cuda.synchronize()
        t_gpu = time.perf_counter() - t1
        assert nit == done
        out["cpu_baseline"] = {
            "value": done / t_used, "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": f"first {done} iterations of the same run at n={n}: oracle/lowpass_oracle.c + oracle/ell_oracle.c, "
                      f"1 thread, {t_used:.1f} s (+{t_init:.1f} s table init); the device runs the same {done} "
                      f"iterations in {t_gpu * 1e3:.1f} ms ({done / t_gpu:.0f} iterations/s), gamma {g2:.6e} vs {cg:.6e}",
            "host_cpus": os.cpu_count(), "gpu_same_sample_iterations_per_s": done / t_gpu,
        }
    print(json.dumps(out), file=real_stdout, flush=True)


def cpu_baseline(n: int, variant: str, kinds, grads, b0, b1, budget_s: float = 20.0):
    """Time the oracle (1 thread, reference loop order incl. the strided mirror stores) on a bounded
    sample of the same cut stream."""
    from oracle import oracle
    cls = oracle.OracleEll if variant == "ell" else oracle.OracleEllStable
    t0 = time.perf_counter()
    if variant == "ell":
        o = cls.new_with_scalar(1.0, np.zeros(n))
    else:  # the same non-trivial factor the GPU space started from (from the identity U and S stay exactly zero)
        from ellalgo_rs_amd import synth
        o = cls.new_with_matrix(1.0, synth.stable_factor(n), np.zeros(n))
    t_init = time.perf_counter() - t0
    done, t_used = 0, 0.0
    while done < len(kinds) and (done < 2 or t_used < budget_s):
        i = done
        t1 = time.perf_counter()
        st = o.update(int(kinds[i]), grads[i], float(b0[i]), None if np.isnan(b1[i]) else float(b1[i]))
        t_used += time.perf_counter() - t1
        assert st == 0, f"oracle cut {i} status {st}"
        done += 1
        if t_used / done * (done + 1) > budget_s and done >= 2:
            break
    extra = None
    if variant == "ell":  # an all-cores line next to the reference's single-threaded loop, labelled as such
        om = cls.new_with_scalar(1.0, np.zeros(n))
        # the CPUs this job may really use (affinity capped by the cgroup quota), unless OMP_NUM_THREADS says otherwise
        nthr = oracle.set_num_threads(0 if os.environ.get("OMP_NUM_THREADS") else oracle.cpu_share())
        d2, t2 = 0, 0.0
        while d2 < len(kinds) and (d2 < 3 or t2 < min(budget_s, 8.0)):
            t1 = time.perf_counter()
            st = om.update_rowwise_mt(int(kinds[d2]), grads[d2], float(b0[d2]), None if np.isnan(b1[d2]) else float(b1[d2]))
            dt = time.perf_counter() - t1
            assert st == 0
            d2 += 1
            if d2 > 1:
                t2 += dt  # the first call pays the OpenMP start-up and the page faults
        extra = {"value": (d2 - 1) / t2, "unit": "updates/s", "threads": nthr,
                 "note": "NOT the reference's loop: row-parallel OpenMP variant of the same arithmetic "
                         "(oracle/ell_oracle.c orc_ell_update_rowwise_mt), bit-identical results"}
    return {
        "all_cores": extra,
        "value": done / t_used,
        "unit": "updates/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{done} updates at n={n} ({variant}), same cut stream, oracle/ell_oracle.c -O3 -ffp-contract=off, "
                  f"{t_used:.1f} s (+{t_init:.1f} s init)",
        "host_cpus": os.cpu_count(),
    }


def gemv_passes(steps: int, dep: int, lookahead: int) -> int:
    """Passes over Q_base a pipelined queue run of `steps` cuts makes at depth `dep` when up to `lookahead` consecutive
    queued cuts share one pass (ELLHIP_OPT_LOOKAHEAD; csrc/ellhip_capi.hip queue_run_multi: a group ends at the apply pass
    and at the end of the run)."""
    i = npend = passes = 0
    while i < steps:
        rem = min(steps - i, dep - npend)
        g = min(lookahead, rem)
        if 3 < lookahead <= 16 and lookahead < rem < 2 * lookahead:
            g = (rem + 1) // 2   # (16-wide passes only: two even groups rather than a full and a small one: group_size)
        passes += 1
        i += g
        npend = (npend + g) % dep
    return passes


def ell_bytes_per_update(n2w: float, sched: str, dep: int, steps: int, symv_mode: bool, lower_apply: bool, sharded: bool = False,
                         lookahead: int = 1):
    """Algorithmic bytes ONE update moves per GPU-share under a schedule / depth IN A TIMED REGION OF `steps` UPDATES
    that starts and ends with nothing recorded (bench.py flushes on both sides), and the model's description.  A
    deferred schedule runs ceil(steps / depth) apply passes inside such a region -- 20 steps at depth 16 hold two -- so
    the per-update figure depends on the region; it tends to the steady-state one for steps >> depth."""
    if dep == 1:
        if sched == "pipelined":
            return 16.0 * n2w, "16*n^2 B/update (rank-1 pass of cut k fused with the GEMV of cut k+1)"
        return 24.0 * n2w, "24*n^2 B/update (GEMV pass 8 + rank-1 pass 16; SURVEY 8d)"
    passes = -(-steps // dep)
    if symv_mode and lower_apply and sched == "pipelined" and lookahead > 1:
        gp = gemv_passes(steps, dep, lookahead)
        per = (4.0 * gp + 8.0 * passes) / steps
        return per * n2w, (f"{per:.4g}*n^2 B/update = ({gp} passes over the lower triangle of 4*n^2, each forming the products of up to "
                           f"{lookahead} QUEUED cuts at once + {passes} lower-triangle apply passes of 8*n^2) / {steps} updates "
                           f"(deferred shrink, depth {dep}, lookahead {lookahead}: only a queue knows the next gradients -- a live "
                           f"cutting-plane loop runs lookahead 1, see host_call_path; steady state "
                           f"{(4.0 * -(-dep // lookahead) + 8.0) / dep:.4g}*n^2; the partial sums the passes hand to the "
                           "reductions, 2.2 n^2 / 64 per vector written and read, are not counted" +
                           ("; per GPU 1/P of that: symmetric row shards of equal trapezoid area, ONE all-reduce of the group's "
                            "n-vectors per pass)" if sharded else ")"))
    if symv_mode and lower_apply:
        per = 4.0 + 8.0 * passes / steps
        return per * n2w, (f"{per:g}*n^2 B/update = ({steps} lower-triangle GEMV passes of 4*n^2 + {passes} lower-triangle apply "
                           f"passes of 8*n^2) / {steps} updates (deferred shrink, depth {dep}; steady state {4.0 + 8.0 / dep:g}*n^2; "
                           "the upper triangle is mirrored back only when Q itself is read" +
                           ("; per GPU 1/P of that: symmetric row shards of equal trapezoid area, one all-reduce of the "
                            "n-vector per update)" if sharded else ")"))
    if symv_mode:
        per = 4.0 + 16.0 * passes / steps
        return per * n2w, (f"{per:g}*n^2 B/update = ({steps} lower-triangle GEMV passes of 4*n^2 + {passes} full apply passes "
                           f"of 16*n^2) / {steps} updates (deferred shrink, depth {dep})")
    if sched == "pipelined":
        per = 8.0 + 8.0 * passes / steps
        return per * n2w, (f"{per:g}*n^2 B/update = ({steps} read-only GEMV passes of 8*n^2, {passes} of them carrying an apply "
                           f"pass: +8*n^2 written) / {steps} updates (deferred shrink, depth {dep})")
    per = 8.0 + 16.0 * passes / steps
    return per * n2w, (f"{per:g}*n^2 B/update = ({steps} read-only GEMV passes of 8*n^2 + {passes} apply passes of 16*n^2) / "
                       f"{steps} updates (deferred shrink, depth {dep})")


# the BASELINE.json configurations besides the headline one, with the brief (steps, warmup, profile steps) they are run
# at inside the default invocation so that the driver's one line carries all of them
BRIEF_CONFIGS = [("n4096-deep", 200, 20, 40), ("n32768-deep", 96, 16, 16), ("n16384-ellstable", 48, 8, 16)]


def symv_kernel_name(pkg, space, symv_mode: bool, n: int, fused: bool = True) -> str:
    look = space.get_option(pkg.capi.OPT_LOOKAHEAD) if (symv_mode and fused and hasattr(space, "get_option")) else 1
    if look > 3 and n % 64 == 0:
        return "k_symm_mfma"
    return "k_symv_multi" if look > 1 else "k_symv"


def run_live_loop(pkg, synth, gen, n: int, kinds, grads, b0, b1, variant: str, steps: int, warm: int = 48):
    """The reference's hot loop `xc() -> oracle -> update_*_cut` (src/cutting_plane.rs:299-311) as an ellalgo-rs user runs it:
    the C++ mirrors of the drivers (host/ellhip/cutting_plane.hpp `cutting_plane_optim`, ell_hip.hpp
    `cutting_plane_optim_pipelined`) over the C ABI, a host oracle that reads the centre (O(n)) and returns synthetic cut k --
    a child process (host/bench/live_loop, plain C++: no Python between the calls), on a fresh handle of the same size.
    The gradient of iteration k + 1 does not exist before update k has returned: no lookahead, no replay."""
    import subprocess
    import tempfile
    need = warm + steps + 1
    if len(kinds) < need:   # (a short run of the queue workloads: the same seeded sequence, longer)
        kinds, grads, b0, b1 = gen(n, need)
    try:
        exe = pkg.build.build_host_tools()
    except Exception as e:   # noqa: BLE001 -- reported, the headline does not depend on it
        return {"skipped": f"host/bench/live_loop could not be built: {e}"}
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=tmpdir) as d:
        path = os.path.join(d, "cuts.bin")
        synth.write_cuts_bin(path, kinds[:need], grads[:need], b0[:need], b1[:need])
        try:
            r = subprocess.run([exe, path, str(warm), str(steps), "ell" if variant == "ell" else "ellstable"],
                               capture_output=True, text=True, timeout=600)
        except subprocess.TimeoutExpired:
            return {"skipped": "host/bench/live_loop timed out"}
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            out = json.loads(line)
            out.pop("case", None)
            out["rc"] = r.returncode
            return out
    return {"skipped": f"host/bench/live_loop printed nothing (rc {r.returncode}): {r.stderr[-300:]}"}


def settle(space) -> None:
    """After a handle with gigabytes of device memory has been destroyed and the next one created, ONE stream synchronisation
    of the new handle within the next few hundred milliseconds takes 75-85 ms longer although the GPU timeline shows its
    kernels back to back (the driver's housekeeping for the recycled memory; measured on this pool with
    tools/probe_brief.py: 88 ms instead of 10.5 for the n = 32768 region right after an n = 16384 handle).  A pause after
    the warm-up absorbs it outside the timed region."""
    time.sleep(0.3)
    space.synchronize()


def brief_config(pkg, synth, torch, workload: str, K: int, W: int, P: int, device: int) -> dict:
    """One BASELINE configuration, timed like the headline run (device-resident queue, pipelined schedule, the depth a
    new handle of that size starts with; recorded updates flushed on both sides of the timed region), reduced to the
    figures `other_configs` carries."""
    n, variant, cutgen, desc = WORKLOADS[workload]
    t_gen = time.perf_counter()
    kinds, grads, b0, b1 = (synth.parallel_cuts if cutgen == "parallel" else synth.deep_cuts)(n, W + K + P)
    if variant == "ell":
        space = pkg.Ell.new_with_scalar(1.0, np.zeros(n), device=device)
        depth = space.defer_depth
    else:
        space = pkg.EllStable.new_with_matrix(1.0, synth.stable_factor(n), np.zeros(n), device=device)
        depth = 1
    log(f"[other-configs] {workload}: inputs ready in {time.perf_counter() - t_gen:.1f}s (depth {depth})")
    fused = variant == "ell"
    space.queue_upload(kinds, grads, b0, b1)
    del grads
    space.queue_run(0, W, fused=fused)
    if variant == "ell":
        space.flush()
    torch.cuda.synchronize()
    space.synchronize()
    settle(space)
    t0 = time.perf_counter()
    space.queue_run(W, K, fused=fused)
    if variant == "ell":
        space.flush()
    space.synchronize()
    elapsed = time.perf_counter() - t0
    space.profile_enable(True)
    space.queue_run(W + K, P, fused=fused)
    space.synchronize()
    prof = space.profile_read()
    space.profile_enable(False)
    status, _ = space.queue_results()
    if not bool(np.all(status == 0)):
        raise SystemExit(f"{workload}: cut {int(np.argmax(status != 0))} did not succeed: benchmark invalid")
    n2 = float(n) * float(n)
    opt = pkg.capi.default_option
    symv_mode = variant == "ell" and n % 2 == 0 and n >= opt(pkg.capi.OPT_SYMV_MIN_N) and opt(pkg.capi.OPT_SYMV) != 0
    lower_apply = symv_mode and opt(pkg.capi.OPT_APPLY_LOWER) != 0
    alg = {"gemv": 8.0 * n2, "rank1": 16.0 * n2, "fused": 16.0 * n2, "apply": (8.0 if lower_apply else 16.0) * n2,
           "apply_gemv": 16.0 * n2, "symv": 4.0 * n2, "stable_fwd": 8.0 * n2, "stable_bwd": 4.0 * n2, "stable_factor": 12.0 * n2}
    resident = variant == "ell" and prof.get("resident", (0.0, 0))[1] > 0
    look = space.get_option(pkg.capi.OPT_LOOKAHEAD) if (variant == "ell" and symv_mode) else 1
    qd = space.get_option(pkg.capi.OPT_QUEUE_DEPTH) if (variant == "ell" and symv_mode) else 0
    dep_eff = qd if (variant == "ell" and symv_mode and lower_apply and look > 3 and n % 64 == 0 and depth == 24 and qd > depth) else depth
    if variant != "ell" and space.get_option(pkg.capi.OPT_STABLE_MIRRORED):
        alg["stable_fwd"], alg["stable_bwd"], alg["stable_factor"] = 4.0 * n2, 4.0 * n2, 0.0
        bytes_update, model = 8.0 * n2, ("8*n^2 B per update (EllStable, mirrored layout: the forward solve reads the factor's upper "
                                         "triangle, the backward solve its mirrored copy, 4 + 4; nothing is rewritten -- the factor update "
                                         "is one running scale per row; the reference moves 24*n^2, the eager kernels 20*n^2)")
    elif variant != "ell":
        fb = 8.0 if space.get_option(pkg.capi.OPT_STABLE_FACTOR) != 0 else 12.0
        alg["stable_factor"] = fb * n2
        if prof.get("stable_factor", (0.0, 0))[1] == 0:   # pulled inside the backward solve's launch
            alg["stable_bwd"] = (4.0 + fb) * n2
        bytes_update, model = (12.0 + fb) * n2, f"{12.0 + fb:g}*n^2 B per update (EllStable: fwd 8 + bwd 4 + factor {fb:g})"
    elif resident:
        # one persistent launch per batch: the lower triangle is parked in the register files (4 n^2 read), written back
        # (4 n^2) and mirrored (16 n^2) ONCE per batch; per update only the gradient (8 n) and O(n) partial sums move
        alg["resident"] = (8.0 * n2 + 8.0 * n * P) if P else 8.0 * n2
        bytes_update = 24.0 * n2 / K + 8.0 * n
        model = (f"(24*n^2 per batch of {K} cuts: park 4 + write back 4 + mirror 16) / {K} + 8*n B per update -- the matrix lives in "
                 "the register files for the whole batch (k_ell_resident): NOT HBM-bound, the update is two on-chip hand-offs "
                 "(one grid barrier) long")
    else:
        bytes_update, model = ell_bytes_per_update(n2, "pipelined", dep_eff, K, symv_mode, lower_apply, False, look)
    per_kernel = {}
    for name, (ms, cnt) in prof.items():
        if cnt:
            e = {"avg_ms": ms / cnt, "launches": cnt}
            if name in alg:
                e["alg_bytes"] = alg[name]
                e["GBps"] = alg[name] / (e["avg_ms"] * 1e-3) / 1e9
            per_kernel[name] = e
    cands = [k for k in per_kernel if k in alg]
    key = (lambda k: per_kernel[k]["avg_ms"] * per_kernel[k]["launches"]) if variant == "ell" else (lambda k: per_kernel[k]["avg_ms"])
    dom = max(cands, key=key) if cands else None
    ms_per_step = elapsed / K * 1e3
    upd_gbps = bytes_update / (ms_per_step * 1e-3) / 1e9
    roofline = {"whole_update": {"alg_bytes": bytes_update, "GBps": upd_gbps, "frac": upd_gbps / HBM_PEAK_GBS, "byte_model": model}}
    if dom:
        kname = {"symv": symv_kernel_name(pkg, space, symv_mode, n), "apply": "k_apply_mfma", "gemv": "k_sweep_gemv_dots",
                 "resident": "k_ell_resident"}.get(dom, dom)
        roofline.update({"kernel": kname if variant == "ell" else "k_st_" + dom[7:], "achieved": per_kernel[dom]["GBps"],
                         "frac": per_kernel[dom]["GBps"] / HBM_PEAK_GBS, "avg_launch_ms": per_kernel[dom]["avg_ms"],
                         "alg_bytes_per_launch": per_kernel[dom]["alg_bytes"]})
    roofline["per_kernel"] = per_kernel
    if resident:
        roofline["bound"] = "on-chip latency (one grid barrier + two L2 hand-offs per update), not HBM"
        roofline["us_per_update_in_launch"] = per_kernel["resident"]["avg_ms"] * 1e3 / max(P, 1)
    del space
    return {"workload": workload, "description": desc, "updates_per_s": K / elapsed, "ms_per_step": ms_per_step, "steps": K,
            "warmup": W, "defer_depth": depth,
            **({"lookahead": look, "queue_depth": dep_eff} if (variant == "ell" and not resident and look > 1) else {}),
            "schedule": ("resident (one launch per batch)" if resident else "pipelined") if variant == "ell" else "ellstable",
            "roofline": roofline}


def main() -> None:
    # The contract is ONE JSON line on stdout.  Native libraries (RCCL prints a version banner) write to
    # fd 1 directly, so park the real stdout and point fd 1 at stderr for the rest of the run.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=os.environ.get("ELLHIP_BENCH_WORKLOAD", "n16384-parallel"),
                    choices=sorted(WORKLOADS) + sorted(LOWPASS_WORKLOADS) + sorted(BATCH_WORKLOADS) + sorted(LMI_WORKLOADS))
    ap.add_argument("--profile-steps", type=int, default=40, help="extra steps with per-kernel HIP events")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--host-path-steps", type=int, default=64,
                    help="extra synchronous ellhip_update() calls from host buffers (PCIe-inclusive rate)")
    ap.add_argument("--live-loop-steps", type=int, default=200,
                    help="iterations of the live loop (xc -> host oracle -> update through the C++ drivers, a child process); "
                         "0 = skip")
    ap.add_argument("--schedule", choices=["pipelined", "two-pass"], default="pipelined",
                    help="pipelined: one pass over Q per update (shrink of cut k fused with the GEMV of cut k+1, "
                         "16*n^2 B); two-pass: GEMV pass + rank-1 pass per update (24*n^2 B). Same results.")
    ap.add_argument("--defer", type=int, choices=[0, 1, 8, 16, 24], default=0,
                    help="0 (default): 16 where the lower-triangle schedule exists (unsharded n even >= 5120, symmetric "
                         "shards), else 8.  "
                         "8: record cuts and apply them to Q in batches of 8 (GEMV passes are read-only, "
                         "8*n^2*(1+1/8) B per update); 1: rewrite Q at every cut like the reference. Same results "
                         "to the 1e-10 parity tolerance.")
    ap.add_argument("--compare-steps", type=int, default=48,
                    help="extra timed steps for each OTHER schedule / depth, reported alongside (0 = skip)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the row-partitioned (multi-GPU) code path even with one rank (rehearsal)")
    ap.add_argument("--other-configs", choices=["auto", "on", "off"], default="auto",
                    help="after the headline run, time the other BASELINE.json configurations briefly in the same process and "
                         "report them as `other_configs` (auto: with the default workload on one GPU)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="ellhip_set_default_option before any handle is created, e.g. --opt APPLY_KERNEL=2 (A/B runs; keys: "
                         "the ELLHIP_OPT_* names of include/ellhip.h without the prefix)")
    ap.add_argument("--all-configs", action="store_true",
                    help="run every BASELINE.json configuration that has a one-GPU form, one after the other (n4096-deep, "
                         "n16384-parallel, n32768-deep, n16384-ellstable), each as its own process with the same "
                         "--steps / --warmup: ONE JSON line of the usual contract per configuration")
    args = ap.parse_args()

    if args.all_configs:
        if args.gpus != 1 or int(os.environ.get("WORLD_SIZE", "1")) != 1:
            raise SystemExit("--all-configs is a one-GPU run")
        import subprocess
        rc = 0
        for wl in ALL_CONFIGS:
            cmd = [sys.executable, os.path.abspath(__file__), "--workload", wl, "--steps", str(args.steps), "--warmup",
                   str(args.warmup), "--profile-steps", str(args.profile_steps), "--compare-steps", str(args.compare_steps),
                   "--host-path-steps", str(args.host_path_steps), "--cpu-budget", str(args.cpu_budget)]
            if args.no_cpu_baseline:
                cmd.append("--no-cpu-baseline")
            log(f"[all-configs] {wl}")
            out = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)   # the child parks its own stdout the same way
            rc = rc or out.returncode
            real_stdout.write(out.stdout)
            real_stdout.flush()
        raise SystemExit(rc)

    if args.workload in LOWPASS_WORKLOADS or args.workload in BATCH_WORKLOADS or args.workload in LMI_WORKLOADS:
        if args.gpus != 1 or int(os.environ.get("WORLD_SIZE", "1")) != 1:
            raise SystemExit(f"{args.workload} is a single-GPU workload (rows f2-f4 of the scope table): run it with --gpus 1")
    if args.workload in LOWPASS_WORKLOADS:
        return lowpass_bench(args, real_stdout)
    if args.workload in BATCH_WORKLOADS:
        return batch_bench(args, real_stdout)
    if args.workload in LMI_WORKLOADS:
        return lmi_bench(args, real_stdout)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    sharded = world > 1 or args.force_sharded
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    import ellalgo_rs_amd as pkg
    from ellalgo_rs_amd import synth

    lib = pkg.capi.load()
    for kv in args.opt:
        key, _, val = kv.partition("=")
        pkg.capi.set_default_option(getattr(pkg.capi, "OPT_" + key.strip().upper()), int(val))
    n, variant, cutgen, desc = WORKLOADS[args.workload]
    K, W, P = args.steps, args.warmup, args.profile_steps
    # EllStable does not shard (its triangular solves are sequential along the partitioned dimension): with N > 1
    # every rank runs an independent replica on its own GPU, no collective in the data path ("replicas only",
    # DESIGN.md section 7); the process group is used for the fences and the max-over-ranks clock only.
    multi = sharded
    replicas = sharded and variant != "ell"
    if replicas:
        sharded = False
    H = args.host_path_steps if not multi else 0
    fused = args.schedule == "pipelined" and variant == "ell"
    depth = args.defer if variant == "ell" else 1
    opt = pkg.capi.default_option
    symv_min_n = opt(pkg.capi.OPT_SYMV_MIN_N)
    lower_ok = n % 2 == 0 and opt(pkg.capi.OPT_SYMV) != 0 and opt(pkg.capi.OPT_APPLY_LOWER) != 0
    if depth == 0:  # auto: 16 pending updates per apply pass where the lower-triangle schedule runs, else 8
        if not sharded:
            depth = 24 if (lower_ok and n >= symv_min_n) else 8
        else:
            # symmetric shards take the pipelined queue run in groups (one pass over the local trapezoid and ONE all-reduce
            # per group of up to 32 queued cuts): the per-cut latency that made small trapezoids a loss is gone
            sym = (n % 64 == 0 and n // 64 >= world and os.environ.get("ELLHIP_SHARD_SYMMETRIC", "1") != "0")
            depth = 24 if (lower_ok and sym) else 8
    C2 = args.compare_steps if variant == "ell" else 0
    # N > 1, depth 8: symmetric row shards (equal lower-trapezoid areas, partial symmetric GEMVs added by one
    # all-reduce, lower-trapezoid apply passes): 5*n^2/P bytes per GPU and update.  That schedule only, so the
    # other schedules are not timed alongside.
    # (taken when a rank's trapezoid still fills the GPU with 64 x 2048 tiles: n = 16384 up to P = 4, n = 32768 up to
    # P = 8; below that the lower-triangle GEMV is latency bound and equal row blocks with full-row GEMVs are faster:
    # measured per rank with tools/shard_timing.py)
    sym_default = "1"
    shard_sym = (sharded and variant == "ell" and depth in (8, 16, 24) and n % 64 == 0 and n // 64 >= world
                 and os.environ.get("ELLHIP_SHARD_SYMMETRIC", sym_default) != "0")
    if shard_sym:
        C2 = 0
    # symmetric shards: a second timed region on the schedule BASELINE config 4 names -- one all-reduce of Q g per ITERATION
    # (the headline region takes the queue in groups: one all-reduce per group of up to 32 queued cuts, replay only)
    C3 = 48 if shard_sym else 0
    # alternatives measured after the main run, on the same handle: (schedule, depth)
    alts = []
    if C2 > 0:
        for alt in (("pipelined", 24), ("pipelined", 16), ("pipelined", 8), ("pipelined", 1), ("two-pass", 8), ("two-pass", 1)):
            if alt[1] >= 16 and (sharded or not lower_ok or n < symv_min_n):
                continue  # depth 16 exists on the lower-triangle schedule only
            if alt != (args.schedule, depth):
                alts.append(alt)
        # the timed configuration with ONE product per pass over Q (what a queue run did before it looked ahead, and the
        # per-cut kernels a live loop's ellhip_update uses): the same handle, ELLHIP_OPT_LOOKAHEAD = 1
        if not sharded and lower_ok and n >= symv_min_n and args.schedule == "pipelined" and depth >= 16:
            alts.insert(0, ("pipelined, lookahead 1", depth))
    # (a second profiled region, kernels one at a time, where the default run overlaps them: see prof_iso below)
    P_iso = P if (variant == "ell" and not sharded and args.schedule == "pipelined" and n % 64 == 0 and n >= symv_min_n and lower_ok) else 0
    # a second, longer region of the SAME schedule when the contract's K is short (the driver runs --steps 20: a 1.25 ms
    # single shot whose one apply pass is a third of it): `config.steady_state_updates_per_s`, never `value`
    S2 = 192 if (variant == "ell" and not sharded and K < 192) else 0
    total = W + K + P + P_iso + 2 * C2 * len(alts) + C3 + S2 + H
    if sharded and n % world and not shard_sym:
        raise SystemExit(f"n={n} is not divisible by {world} ranks")

    t_gen = time.perf_counter()
    kinds, grads, b0, b1 = (synth.parallel_cuts if cutgen == "parallel" else synth.deep_cuts)(n, total)
    log(f"[rank {rank}] generated {total} cuts for {args.workload} in {time.perf_counter() - t_gen:.1f}s")

    # ---- build the search space (Q0 = I, xc0 = 0, kappa0 = 1)
    nrows = n // world
    row0 = rank * nrows
    if not sharded and variant != "ell":
        # EllStable starts from a NON-trivial packed state (random unit-upper-triangular factor, random positive
        # diagonal, junk in the scratch triangle; EllStable::new_with_matrix, src/ell_stable.rs:18-27): from the
        # identity the bug-compatible arithmetic keeps the factor and the scratch triangle exactly zero
        # and the solves would stream zeros.
        space = pkg.EllStable.new_with_matrix(1.0, synth.stable_factor(n), np.zeros(n), device=local_rank)
    elif not sharded:
        space = pkg.Ell.new_with_scalar(1.0, np.zeros(n), device=local_rank)
    else:
        # The row-partitioned space behind the C ABI (include/ellhip_sharded.h): libellhip.so issues the one collective
        # of an update through RCCL itself and the pipelined loop over the queue runs in C.  torch.distributed only
        # carries the communicator's unique id to the ranks (and the fences / the max-over-ranks clock below).
        # ELLHIP_BENCH_SHARDED=torch selects the older orchestration of the same shard handle through
        # torch.distributed collectives (ellalgo-rs_amd/sharded.py); it is also the fallback if RCCL cannot be opened.
        space = None
        if os.environ.get("ELLHIP_BENCH_SHARDED", "abi") == "abi":
            from ellalgo_rs_amd import sharded_abi
            # The choice between the C-ABI space and the torch.distributed fallback is COLLECTIVE: rank 0 always
            # broadcasts (the id, or None if RCCL could not be opened), every rank then tries to create its handle and the
            # ranks agree (all-reduce MIN of an ok flag) -- a rank that failed alone must not leave the others inside
            # RCCL collectives it never joins.
            ids = [None]
            if rank == 0:
                try:
                    ids[0] = sharded_abi.unique_id()
                except (pkg.capi.EllHipError, RuntimeError, OSError) as e:
                    log(f"[rank 0] ellhip_sharded_unique_id failed ({type(e).__name__}: {e})")
            dist.broadcast_object_list(ids, src=0)
            ok_local = 0
            if ids[0] is not None:
                try:
                    space = pkg.ShardedEllAbi.new_with_scalar(1.0, np.zeros(n), device=local_rank, rank=rank, nranks=world,
                                                              nccl_id=ids[0], symmetric=shard_sym,
                                                              defer_depth=depth if shard_sym else 8)
                    ok_local = 1
                except (pkg.capi.EllHipError, RuntimeError, OSError) as e:
                    log(f"[rank {rank}] C-ABI sharded space unavailable ({type(e).__name__}: {e})")
            flag = torch.tensor([ok_local], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                sharded_via = "c-abi (RCCL inside libellhip.so)"
            else:
                if space is not None:
                    del space
                space = None
                log(f"[rank {rank}] falling back to torch.distributed collectives on every rank")
        if space is None:
            from ellalgo_rs_amd.sharded import ShardedEll
            space = ShardedEll.new_with_scalar(1.0, np.zeros(n), device=local_rank, symmetric=shard_sym,
                                              defer_depth=depth if shard_sym else 8)
            sharded_via = "torch.distributed"
    nq = W + K + P + P_iso + 2 * C2 * len(alts) + C3 + S2
    if variant == "ell" and not shard_sym:   # always explicit: a new unsharded handle may start at depth 16 by itself
        space.set_defer_depth(depth) if sharded else setattr(space, "defer_depth", depth)
    space.queue_upload(kinds[:nq], grads[:nq], b0[:nq], b1[:nq])

    def run(first: int, count: int, use_fused: bool = fused) -> None:
        space.queue_run(first, count, fused=use_fused)

    def fence() -> None:
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- warm-up, then the timed region: EXACTLY K steps between two fences
    # The deferred schedule applies its recorded updates every `depth` cuts.  So that the timed region pays for ALL
    # of its K updates whatever K and W are, the recorded ones are applied before it starts (untimed) and again at
    # its end (timed): ceil(K / depth) apply passes inside the region, never fewer.
    run(0, W)
    if variant == "ell":
        space.flush()
    fence()
    time.sleep(0.3)   # (see settle(): driver housekeeping after large allocations, kept out of the timed region)
    fence()
    t0 = time.perf_counter()
    run(W, K)
    if variant == "ell":
        space.flush()
    fence()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel durations with HIP events on the launch stream (outside the timed region)
    prof = None
    if P > 0:
        space.profile_enable(True)
        run(W + K, P)
        space.synchronize()
        prof = space.profile_read()
        space.profile_enable(False)
    # ... and the same kernels one at a time: with ELLHIP_OPT_OVERLAP the next group's products run on a second stream
    # beside this group's stage, which stretches both (they share the CUs and HBM); the isolated durations say what each
    # kernel does by itself.  `roofline.frac` stays the as-run figure (what rocprofv3 sees for this command).
    prof_iso = None
    if P_iso > 0 and fused and space.get_option(pkg.capi.OPT_OVERLAP) != 0 and space.get_option(pkg.capi.OPT_LOOKAHEAD) > 3:
        space.set_option(pkg.capi.OPT_OVERLAP, 0)
        space.profile_enable(True)
        run(W + K + P, P)
        space.synchronize()
        prof_iso = space.profile_read()
        space.profile_enable(False)
        space.set_option(pkg.capi.OPT_OVERLAP, 1)
    elif P_iso > 0:
        run(W + K + P, P_iso)   # (nothing overlaps in this configuration: the cuts set aside for the second region just run)

    # ---- the other schedules / depths on the same handle, for comparison (timed the same way + per-kernel events)
    others = []
    pos = W + K + P + P_iso
    for (alt_sched, alt_depth) in alts:
        alt_fused = alt_sched.startswith("pipelined")
        space.set_defer_depth(alt_depth) if sharded else setattr(space, "defer_depth", alt_depth)
        look1 = alt_sched.endswith("lookahead 1")
        if look1:
            look_was = space.get_option(pkg.capi.OPT_LOOKAHEAD)
            space.set_option(pkg.capi.OPT_LOOKAHEAD, 1)
        fence()
        t2 = time.perf_counter()
        run(pos, C2, alt_fused)
        space.flush()  # (no-op when C2 is a multiple of the depth: the region always pays for all of its updates)
        fence()
        el2 = time.perf_counter() - t2
        if sharded:
            t = torch.tensor([el2], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el2 = float(t.item())
        space.profile_enable(True)
        run(pos + C2, C2, alt_fused)
        space.synchronize()
        others.append({"schedule": alt_sched, "defer_depth": alt_depth, "steps": C2, "updates_per_s": C2 / el2,
                       "ms_per_step": el2 / C2 * 1e3, "prof": space.profile_read()})
        space.profile_enable(False)
        if look1:
            space.set_option(pkg.capi.OPT_LOOKAHEAD, look_was)
        pos += 2 * C2

    per_iter = None
    if C3 > 0 and hasattr(space, "set_local_option"):
        look_was = space.get_local_option(pkg.capi.OPT_LOOKAHEAD)
        space.set_local_option(pkg.capi.OPT_LOOKAHEAD, 1)
        space.flush()
        fence()
        t2 = time.perf_counter()
        run(pos, C3, True)
        space.flush()
        fence()
        el3 = time.perf_counter() - t2
        t = torch.tensor([el3], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el3 = float(t.item())
        space.set_local_option(pkg.capi.OPT_LOOKAHEAD, look_was)
        per_iter = {"steps": C3, "updates_per_s": C3 / el3, "ms_per_step": el3 / C3 * 1e3, "collectives_per_update": 1.0,
                    "schedule": "pipelined, one all-reduce(Q g) of n doubles per update (ELLHIP_OPT_LOOKAHEAD = 1 on the shard handle): "
                                "the form BASELINE config 4 names; what a live cutting-plane loop over a sharded space can reach"}
        pos += C3

    steady = None
    if S2 > 0:
        space.defer_depth = depth
        space.flush()
        fence()
        t2 = time.perf_counter()
        run(pos, S2)
        space.flush()
        fence()
        steady = S2 / (time.perf_counter() - t2)
        pos += S2

    status, tsqs = space.queue_results()
    ran = nq if per_iter or C3 == 0 else nq - C3
    ok = bool(np.all(status[:ran] == 0))
    if not ok:
        bad = int(np.argmax(status[:ran] != 0))
        raise SystemExit(f"cut {bad} did not succeed (status {int(status[bad])}): benchmark invalid")
    if cutgen == "parallel":
        # the true parallel branch must have been taken: tsq > beta1^2 (src/ell_calc.rs:761,840)
        assert bool(np.all(tsqs[:ran] > b1[:ran] ** 2)), "a parallel cut fell back to a single cut"

    # ---- synchronous host-buffer path (PCIe-inclusive), N = 1 only
    host_path = None
    if H > 0:
        # the drop-in SearchSpace call as a Rust binding makes it: once with the reference's data flow (depth 1) and
        # once with the timed configuration's depth; the second half of the calls at each depth is timed
        host_path = {"steps": H, "note": "ellhip_update() per call: pageable host gradient in over PCIe, status out, "
                                         "synchronous at return; never used as `value`"}
        i = nq
        for dep in ((1, depth) if (variant == "ell" and depth != 1) else (1,)):
            if variant == "ell":
                space.defer_depth = dep
            per = H // (2 if (variant == "ell" and depth != 1) else 1)
            warm = per // 2
            for j in range(per):
                if j == warm:
                    space.flush() if variant == "ell" else None
                    space.synchronize()
                    t2 = time.perf_counter()
                st = space._update(int(kinds[i]), (grads[i], (b0[i], None if np.isnan(b1[i]) else b1[i])))
                assert int(st) == 0
                i += 1
            space.flush() if variant == "ell" else None
            space.synchronize()
            rate = (per - warm) / (time.perf_counter() - t2)
            host_path["updates_per_s" if dep == 1 else f"updates_per_s_depth{dep}"] = rate

    # ---- the live loop: xc() -> host oracle -> update, through the C++ drivers over the C ABI (N = 1 only)
    live_loop = None
    if H > 0 and args.live_loop_steps > 0:
        live_loop = run_live_loop(pkg, synth, synth.parallel_cuts if cutgen == "parallel" else synth.deep_cuts, n, kinds, grads,
                                  b0, b1, variant, args.live_loop_steps)

    if rank != 0:
        dist.barrier()
        dist.destroy_process_group()
        return

    n2w = float(n) * float(n) / (1 if replicas else world)  # n^2 per GPU
    ms_per_step = elapsed / K * 1e3
    value = K / elapsed * (world if replicas else 1)  # replicas: every rank completed K updates of its own
    # algorithmic bytes per launch of each kernel class (per GPU)
    symv_mode = ((not sharded) and n % 2 == 0 and n >= symv_min_n and opt(pkg.capi.OPT_SYMV) != 0) or shard_sym
    lower_apply = symv_mode and opt(pkg.capi.OPT_APPLY_LOWER) != 0
    alg = {"gemv": 8.0 * n2w, "rank1": 16.0 * n2w, "fused": 16.0 * n2w, "apply": (8.0 if lower_apply else 16.0) * n2w,
           "apply_gemv": 16.0 * n2w, "symv": 4.0 * n2w,
           "stable_fwd": 8.0 * n * n, "stable_bwd": 4.0 * n * n, "stable_factor": 12.0 * n * n,
           "resident": 8.0 * n2w}   # one launch per batch: park 4 n^2 + write back 4 n^2 (+ 8 n per cut)
    if variant != "ell":
        # EllStable: the factor update rewrites the strict upper triangle from itself (8*n^2; the scratch entry the
        # reference adds IS fl(U*w)) unless the scratch-reading tile kernel is forced; and it runs
        # inside the backward solve's launch (k_st_bwd_factor) when no separate factor launch shows up in the profile
        rows = space.get_option(pkg.capi.OPT_STABLE_FACTOR) != 0
        alg["stable_factor"] = (8.0 if rows else 12.0) * n * n
        if prof and prof.get("stable_factor", (0.0, 0))[1] == 0:
            alg["stable_bwd"] = 4.0 * n * n + alg["stable_factor"]
        if space.get_option(pkg.capi.OPT_STABLE_MIRRORED):
            # mirrored layout: each solve reads its triangle of the (never rewritten) base factor once, 4 n^2, and stores
            # nothing but vectors: no scratch triangle, no factor pass (the factor update is one running scale per row)
            alg["stable_fwd"], alg["stable_bwd"], alg["stable_factor"] = 4.0 * n * n, 4.0 * n * n, 0.0

    lookahead, queue_depth = 1, 0
    if variant == "ell" and symv_mode and not sharded:
        lookahead = space.get_option(pkg.capi.OPT_LOOKAHEAD)
        queue_depth = space.get_option(pkg.capi.OPT_QUEUE_DEPTH)
    elif shard_sym and str(sharded_via).startswith("c-abi"):
        # symmetric shards behind the C ABI take the pipelined queue run in groups too (ellhip_sharded_queue_run_fused)
        lookahead = opt(pkg.capi.OPT_LOOKAHEAD)
        queue_depth = opt(pkg.capi.OPT_QUEUE_DEPTH)

    def eff_depth(sched, dep):
        """Recorded updates per apply pass: inside a pipelined queue run on the group stage (lookahead > 3, n % 64 == 0) a
        depth-24 handle lets them pile up to ELLHIP_OPT_QUEUE_DEPTH."""
        deep = sched == "pipelined" and dep == 24 and lookahead > 3 and n % 64 == 0 and lower_apply and queue_depth > dep
        return queue_depth if deep else dep

    def byte_model(sched, dep, steps):
        return ell_bytes_per_update(n2w, sched, eff_depth(sched, dep), steps, symv_mode, lower_apply, sharded, lookahead)

    def kernel_table(pr):
        tab = {}
        for name, (ms, cnt) in (pr or {}).items():
            if cnt:
                avg_ms = ms / cnt
                e = {"avg_ms": avg_ms, "launches": cnt}
                if name in alg:
                    e["alg_bytes"] = alg[name]
                    e["GBps"] = alg[name] / (avg_ms * 1e-3) / 1e9
                tab[name] = e
        return tab

    per_kernel = kernel_table(prof)
    # bytes one update moves under the schedule that was timed (each schedule has its OWN byte model;
    # nothing is credited against the 24*n^2 two-pass model)
    resident_run = variant == "ell" and bool(prof) and prof.get("resident", (0.0, 0))[1] > 0
    if variant != "ell" and space.get_option(pkg.capi.OPT_STABLE_MIRRORED):
        bytes_update, model = 8.0 * n * n, ("8*n^2 B per update (EllStable, mirrored layout: the forward solve reads the factor's upper "
                                            "triangle, the backward solve its mirrored copy, 4 + 4; nothing is rewritten -- the factor "
                                            "update is one running scale per row; the reference moves 24*n^2, the eager kernels 20*n^2)")
    elif variant != "ell":
        fb = alg["stable_factor"] / (n * n)
        bytes_update, model = (12.0 + fb) * n * n, f"{12.0 + fb:g}*n^2 B per update (EllStable: fwd 8 + bwd 4 + factor {fb:g})"
    elif resident_run:
        bytes_update = 24.0 * n2w / K + 8.0 * n
        model = (f"(24*n^2 per batch of {K} cuts: park 4 + write back 4 + mirror 16) / {K} + 8*n B per update -- the matrix lives in "
                 "the register files for the whole batch (k_ell_resident): NOT HBM-bound, the update is two on-chip hand-offs "
                 "(one grid barrier) long")
    else:
        bytes_update, model = byte_model("pipelined" if fused else "two-pass", depth, K)
    roofline = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None, "byte_model": model}
    if variant == "ell":
        # dominant = the kernel class with the largest total time in the profiled steps
        cands = [k for k in per_kernel if k in alg]
        dom = max(cands, key=lambda k: per_kernel[k]["avg_ms"] * per_kernel[k]["launches"]) if cands else None
    else:
        cands = [k for k in per_kernel if k in alg]
        dom = max(cands, key=lambda k: per_kernel[k]["avg_ms"]) if cands else None
    if dom in per_kernel:
        kname = {"symv": ("k_symm_mfma" if lookahead > 3 and n % 64 == 0 else "k_symv_multi") if (lookahead > 1 and fused) else "k_symv",
                 "apply": "k_apply_mfma" if depth == 24 else "k_apply_lower", "apply_gemv": "k_sweep_apply",
                 "resident": "k_ell_resident"}.get(dom, "k_sweep:" + dom)
        roofline.update({"kernel": kname if variant == "ell" else "k_st_" + dom[7:],
                         "achieved": per_kernel[dom]["GBps"], "frac": per_kernel[dom]["GBps"] / HBM_PEAK_GBS,
                         "alg_bytes_per_launch": per_kernel[dom]["alg_bytes"],
                         "avg_launch_ms": per_kernel[dom]["avg_ms"]})
    try:  # measured HBM traffic per launch (PMC counters, collected separately under rocprofv3)
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f).get(args.workload)
        if pmc and world == 1 and dom:
            roofline["traffic"] = pmc.get(dom)
            roofline["traffic_source"] = pmc.get("source")
            # the figure comes from separate rocprofv3 --pmc passes of the same command (profiles/), not from this run:
            # what THIS run measured live is the per-kernel HIP-event table below (`per_kernel`)
            roofline["traffic_measured_in_this_run"] = False
    except OSError:
        pass
    upd_gbps = bytes_update / (ms_per_step * 1e-3) / 1e9
    roofline["per_kernel"] = per_kernel
    symm_width = None
    if roofline.get("kernel") == "k_symm_mfma":
        # The profiled window's passes carry P / launches gradients on average (groups of 32 and 16 at the default lookahead): above
        # the ridge the binding roof is the FP64 matrix pipe and `roofline` says so (bound "mfma", TFLOP/s of the gradients actually
        # multiplied); the HBM view of the same launches stays beside it as `roofline.hbm_view` (product_pass_roof).
        symm_width = P / max(per_kernel[dom]["launches"], 1)
        roof = product_pass_roof(n2w, symm_width, roofline["avg_launch_ms"])
        roofline["matrix_pipe"] = {"flop_per_launch": roof["flop_per_launch"], "gradients_per_pass": symm_width,
                                   "peak_TFLOPs": FP64_MFMA_PEAK_TF, "achieved_TFLOPs": roof["TFLOPs"], "frac": roof["mfma_frac"],
                                   "note": "HBM time and matrix-pipe time of this kernel add up rather than overlap "
                                           "(DESIGN.md 3.6): read the HBM fraction and matrix_pipe.frac together"}
        if roof["bound"] == "mfma":
            roofline["hbm_view"] = {"achieved": roofline["achieved"], "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": roofline["frac"],
                                    "alg_bytes_per_launch": roofline["alg_bytes_per_launch"]}
            roofline.update({"bound": "mfma", "unit": "TFLOP/s", "peak": FP64_MFMA_PEAK_TF, "achieved": roof["TFLOPs"],
                             "frac": roof["mfma_frac"], "alg_flop_per_launch": roof["flop_per_launch"],
                             "bound_note": f"{symm_width:g} gradients per pass = {roof['flop_per_byte']:g} flop/B, above this card's "
                                           f"ridge ({roof['ridge']:.1f}): priced against the FP64 matrix pipe; hbm_view has the same "
                                           f"launches against HBM"})
    if prof_iso:
        iso = kernel_table(prof_iso)
        roofline["per_kernel_isolated"] = iso
        if dom in iso and "GBps" in iso[dom]:
            roofline["isolated"] = {"kernel": roofline.get("kernel"), "avg_launch_ms": iso[dom]["avg_ms"], "achieved": iso[dom]["GBps"],
                                    "frac": iso[dom]["GBps"] / HBM_PEAK_GBS, "unit": "GB/s",
                                    "note": "the same kernel with nothing running beside it (ELLHIP_OPT_OVERLAP = 0 for these launches only)"}
            if roofline.get("bound") == "mfma" and symm_width:
                w_iso = P_iso / max(iso[dom]["launches"], 1)
                roof_iso = product_pass_roof(n2w, w_iso, iso[dom]["avg_ms"])
                roofline["isolated"].update({"hbm_frac": roofline["isolated"]["frac"], "achieved": roof_iso["TFLOPs"], "unit": "TFLOP/s",
                                             "frac": roof_iso["mfma_frac"], "gradients_per_pass": w_iso})
    roofline["whole_update"] = {"alg_bytes_per_gpu": bytes_update, "GBps_per_gpu": upd_gbps,
                                "frac": upd_gbps / HBM_PEAK_GBS}
    if "achieved" not in roofline:
        roofline.update({"kernel": "whole_update", "achieved": upd_gbps, "frac": upd_gbps / HBM_PEAK_GBS})
    for o in others:
        if o["schedule"].endswith("lookahead 1"):
            ob, omodel = ell_bytes_per_update(n2w, "pipelined", o["defer_depth"], o["steps"], symv_mode, lower_apply, sharded, 1)
        else:
            ob, omodel = byte_model(o["schedule"], o["defer_depth"], o["steps"])
        og = ob / (o["ms_per_step"] * 1e-3) / 1e9
        o["byte_model"] = omodel
        o["whole_update"] = {"alg_bytes_per_gpu": ob, "GBps_per_gpu": og, "frac": og / HBM_PEAK_GBS}
        o["per_kernel"] = kernel_table(o.pop("prof"))

    out = {
        "metric": "ellipsoid updates/sec at n=%d; achieved HBM GB/s vs peak" % n,
        "value": value,
        "unit": "updates/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak" if replicas else "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": args.workload, "n": n, "space": variant, "cuts": cutgen,
                   "schedule": (("resident (one launch per batch)" if resident_run else ("pipelined" if fused else "two-pass"))
                                if variant == "ell" else "ellstable"),
                   "defer_depth": depth,
                   **({"lookahead": lookahead, "queue_depth": eff_depth("pipelined", depth),
                       "lookahead_note": "products of up to `lookahead` consecutive QUEUED cuts per pass over Q (ELLHIP_OPT_LOOKAHEAD); "
                                         "a live cutting-plane loop cannot look ahead: see host_call_path and other_schedules"}
                      if (variant == "ell" and fused and lookahead > 1 and not resident_run) else {}),
                   "description": desc,
                   "partition": (f"symmetric row shards x{world} (boundaries at n*sqrt(r/P), all-reduce)" if shard_sym
                                 else f"row-block x{world} (all-gather)") if sharded
                                else (f"replicas only x{world} (independent search spaces, no data-path collective)"
                                      if replicas else "none"),
                   "q_bytes_per_gpu": 8.0 * n * n / (1 if replicas else world),
                   **({"options": args.opt} if args.opt else {}),
                   **({"collective_issued_by": sharded_via} if sharded else {})},
        "roofline": roofline,
    }
    if sharded and variant == "ell":
        grp = shard_sym and lookahead > 3
        cpu_ = (gemv_passes(K, eff_depth("pipelined", depth), lookahead) / K) if grp else 1.0
        out["config"]["collectives_per_update"] = cpu_
        out["config"]["collective_payload_doubles"] = (n * min(lookahead, 16)) if grp else (n if shard_sym else n // world)
        if per_iter:
            out["per_iteration_collective"] = per_iter
            out["config"]["per_iteration_collective_updates_per_s"] = per_iter["updates_per_s"]
        elif not grp:
            out["config"]["per_iteration_collective_updates_per_s"] = value   # (the headline schedule already is that form)
    if others:
        out["other_schedules"] = others
    if host_path:
        out["host_call_path"] = host_path
    if live_loop:
        out["live_loop"] = live_loop
    # What a SearchSpace caller can reach, copied into the two objects a record keeper is sure to keep (`config`, `roofline`):
    # the headline `value` replays a QUEUE and forms the products of up to `lookahead` FUTURE gradients per pass over Q, which a
    # live cutting-plane loop can never supply (src/cutting_plane.rs:299-311: gradient k + 1 comes from the oracle at the centre
    # update k produced).
    reach = {}
    if variant == "ell" and fused and lookahead > 1 and not resident_run:
        reach["value_requires_future_gradients"] = lookahead
    elif resident_run:
        reach["value_requires_a_queued_batch"] = K
    if live_loop and "plain_iterations_per_s" in live_loop:
        reach["live_loop_updates_per_s"] = live_loop["plain_iterations_per_s"]
        reach["live_loop_pipelined_updates_per_s"] = live_loop["pipelined_iterations_per_s"]
    if host_path:
        key = f"updates_per_s_depth{depth}" if (variant == "ell" and depth != 1) else "updates_per_s"
        if key in host_path:
            reach["host_call_updates_per_s"] = host_path[key]
    for o in others:
        if o["schedule"].endswith("lookahead 1"):
            reach["lookahead1_updates_per_s"] = o["updates_per_s"]
        if o["schedule"] == "two-pass" and o["defer_depth"] == 1:
            roofline["canonical_24n2_updates_per_s"] = o["updates_per_s"]
            roofline["canonical_24n2_frac"] = o["whole_update"]["frac"]
    if steady:
        reach["steady_state_updates_per_s"] = steady
        reach["steady_state_note"] = (f"the timed schedule over {S2} more steps of the same queue (same fences, recorded updates applied inside): "
                                      f"`value` is the contract's {K}-step region, {ms_per_step * K:.2f} ms long, whose closing apply pass "
                                      "is a large part of it")
    out["config"].update(reach)
    if world == 1 and not args.no_cpu_baseline:
        log("[rank 0] timing the CPU oracle (bounded sample) ...")
        out["cpu_baseline"] = cpu_baseline(n, variant, kinds, grads, b0, b1, args.cpu_budget)
    want_others = args.other_configs == "on" or (args.other_configs == "auto" and args.workload == "n16384-parallel")
    if world == 1 and not args.force_sharded and want_others:
        # BASELINE.json's other configurations (2, 4 in its one-GPU form, 5), briefly, on the same process and GPU
        del space, grads
        out["other_configs"] = [brief_config(pkg, synth, torch, wl, k, w, pr, local_rank) for (wl, k, w, pr) in BRIEF_CONFIGS
                                if wl != args.workload]
    print(json.dumps(out), file=real_stdout, flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
