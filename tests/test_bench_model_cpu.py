"""CPU: bench.py's byte models (no GPU, no engine): the passes over Q a pipelined queue run makes with the queue's
lookahead (csrc/ellhip_capi.hip queue_run_multi groups the cuts the same way) and the per-update byte counts the JSON line
carries.  Reference quantities: one update = src/ell.rs:97-137 (GEMV 8 n^2 + rank-1 16 n^2 bytes in its own data flow)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_model", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_passes_over_q_with_lookahead():
    g = bench.gemv_passes
    assert g(20, 48, 16) == 2          # the driver's 20 steps: two even groups of 10
    assert g(16, 48, 16) == 1
    assert g(17, 48, 16) == 2          # 9 + 8
    assert g(48, 48, 16) == 3
    assert g(200, 48, 16) == 13        # 4 x (16 16 16) + 8
    assert g(96, 48, 16) == 6
    assert g(64, 24, 16) == 5          # depth 24 without the deeper queue: (12 12) (12 12) (16)
    assert g(24, 24, 16) == 2          # 12 + 12 rather than 16 + 8
    assert g(100, 24, 1) == 100        # one product per pass
    assert g(10, 24, 3) == 4           # vector-ALU groups: 3 3 3 1 (no even split below lookahead 4)
    # lookahead 32 (the default): groups of 17 .. 32 ride on one pass with two column tiles
    assert g(20, 48, 32) == 1          # the driver's 20 steps: one pass
    assert g(48, 48, 32) == 2          # 32 + 16
    assert g(200, 48, 32) == 9         # 4 x (32 16) + 8
    assert g(96, 48, 32) == 4
    assert g(33, 48, 32) == 2          # 32 + 1


def test_bytes_per_update():
    n2 = 16384.0 ** 2
    b, _ = bench.ell_bytes_per_update(n2, "two-pass", 1, 100, True, True)
    assert b == 24.0 * n2              # the reference's data flow
    b, _ = bench.ell_bytes_per_update(n2, "pipelined", 1, 100, True, True)
    assert b == 16.0 * n2
    b, _ = bench.ell_bytes_per_update(n2, "pipelined", 24, 48, True, True)
    assert abs(b - (4.0 + 8.0 * 2 / 48) * n2) < 1.0            # one product per pass, two apply passes
    b, text = bench.ell_bytes_per_update(n2, "pipelined", 48, 200, True, True, False, 16)
    assert abs(b - (13 * 4.0 + 5 * 8.0) / 200 * n2) < 1.0 and "lookahead 16" in text
    b, _ = bench.ell_bytes_per_update(n2, "pipelined", 24, 20, True, True, False, 16)
    assert abs(b - (2 * 4.0 + 1 * 8.0) / 20 * n2) < 1.0        # the driver's form: 0.8 n^2 per update
    b, text = bench.ell_bytes_per_update(n2 / 4, "pipelined", 48, 96, True, True, True, 16)
    assert abs(b - (6 * 4.0 + 2 * 8.0) / 96 * n2 / 4) < 1.0 and "all-reduce" in text   # symmetric shards, per GPU


def test_product_pass_roof_crosses_the_ridge_at_twenty_gradients():
    """bench.py prices the matrix-core product pass against the FP64 matrix pipe when a pass carries enough gradients to sit above
    the card's ridge (2 n^2 flop per gradient for 4 n^2 bytes: gradients / 2 flop per byte against 78.6 TFLOP/s / 8 TB/s = 9.8),
    against HBM otherwise; the numbers of profiles/r04 as the example."""
    n2 = 16384.0 ** 2
    r16 = bench.product_pass_roof(n2, 16, 0.27)
    assert r16["bound"] == "hbm" and abs(r16["flop_per_byte"] - 8.0) < 1e-12 and abs(r16["hbm_frac"] - 4 * n2 / 0.27e-3 / 8e12) < 1e-12
    r19, r20 = bench.product_pass_roof(n2, 19.6, 0.4), bench.product_pass_roof(n2, 19.7, 0.4)
    assert r19["bound"] == "hbm" and r20["bound"] == "mfma" and abs(r20["ridge"] - 9.825) < 1e-9
    r32 = bench.product_pass_roof(n2, 32, 0.445)
    assert r32["bound"] == "mfma" and abs(r32["TFLOPs"] - 2 * n2 * 32 / 0.445e-3 / 1e12) < 1e-9 and 0.48 < r32["mfma_frac"] < 0.50
    assert abs(r32["hbm_frac"] - 0.3016) < 1e-3      # the same launch against HBM: 4 n^2 bytes in 0.445 ms
    # a row shard: a quarter of the elements, a quarter of the flop and of the bytes, the same intensity
    assert bench.product_pass_roof(n2 / 4, 32, 0.2)["flop_per_byte"] == 16.0
