"""Synthetic cut sequences for the benchmark configurations.

Deterministic and self-contained: a splitmix64 counter stream + Box-Muller written here in numpy
(no dependence on a library's Gaussian sampler), so the CPU oracle, the GPU engine and every rank of
a multi-GPU run see bit-identical inputs.
"""
from __future__ import annotations

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(seed: int, count: int) -> np.ndarray:
    """count 64-bit outputs of splitmix64 started at `seed` (vectorised: counter based)."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + np.arange(1, count + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def uniform01(seed: int, count: int) -> np.ndarray:
    """Uniforms in (0, 1): top 53 bits, offset by half an ulp so 0 never occurs."""
    return ((_splitmix64(seed, count) >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def unit_gaussian(seed: int, n: int) -> np.ndarray:
    """n iid N(0,1) draws (Box-Muller) normalised to unit 2-norm."""
    m = (n + 1) // 2
    u = uniform01(seed, 2 * m)
    r = np.sqrt(-2.0 * np.log(u[:m]))
    t = 2.0 * np.pi * u[m:]
    g = np.empty(2 * m, dtype=np.float64)
    g[0::2] = r * np.cos(t)
    g[1::2] = r * np.sin(t)
    g = g[:n]
    return g / np.sqrt(np.dot(g, g))


SEED0 = 0x5EED0000


def deep_cuts(n: int, k: int, seed0: int = SEED0):
    """Config 2/4/5: update_bias_cut(SingleCut(beta_k)), beta_k ~ U[0, 0.1)."""
    grads = np.empty((k, n), dtype=np.float64)
    for i in range(k):
        grads[i] = unit_gaussian(seed0 + i, n)
    beta0 = 0.1 * uniform01(seed0 ^ 0xBE7A, k)
    # the specified stream is 20 warm-up + 200 timed cuts; cuts drawn beyond those (per-kernel event
    # pass, host-call pass) use a 5x shallower range so they still succeed once tau has shrunk
    beta0[220:] *= 0.2
    kinds = np.zeros(k, dtype=np.int32)  # CUT_BIAS
    beta1 = np.full(k, np.nan)
    return kinds, grads, beta0, beta1


def parallel_cuts(n: int, k: int, seed0: int = SEED0):
    """Config 3: alternate update_central_cut(ParallelCut(0, Some(b1))), b1 ~ U[0.05, 0.5) and
    update_bias_cut(ParallelCut(b0, Some(b1))), b0 ~ U[0, 0.05), b1 = b0 + U[0.05, 0.45)."""
    grads = np.empty((k, n), dtype=np.float64)
    for i in range(k):
        grads[i] = unit_gaussian(seed0 + i, n)
    u = uniform01(seed0 ^ 0xBE7A, 2 * k)
    kinds = np.empty(k, dtype=np.int32)
    beta0 = np.empty(k, dtype=np.float64)
    beta1 = np.empty(k, dtype=np.float64)
    for i in range(k):
        if i % 2 == 0:
            kinds[i] = 1  # CUT_CENTRAL
            beta0[i] = 0.0
            beta1[i] = 0.05 + 0.45 * u[2 * i]
        else:
            kinds[i] = 0  # CUT_BIAS
            beta0[i] = 0.05 * u[2 * i]
            beta1[i] = beta0[i] + 0.05 + 0.40 * u[2 * i + 1]
    return kinds, grads, beta0, beta1


def stable_factor(n: int, seed: int = 0x5EEDFAC7, chunk_rows: int = 1024) -> np.ndarray:
    """A non-trivial packed EllStable state for `EllStable::new_with_matrix` (src/ell_stable.rs:18-27): one n x n
    row-major buffer with a positive diagonal d ~ U[0.5, 1.5), a unit-upper-triangular factor whose strict upper
    entries are N(0, 1) * 0.1 / sqrt(n) (so every row of L stays O(0.1) in norm) and a strict lower (scratch)
    triangle filled with junk of the same size -- the reference overwrites every scratch entry in the forward solve
    before anything reads it (src/ell_stable.rs:61-69), so the junk must never show in a result.  With the identity
    factor of `new_with_scalar` the strict upper and scratch triangles stay exactly zero forever, which
    exercises none of the triangular-solve arithmetic; this factor does."""
    m = np.empty((n, n), dtype=np.float64)
    scale = 0.1 / np.sqrt(float(n))
    for r0 in range(0, n, chunk_rows):
        rows = min(chunk_rows, n - r0)
        cnt = rows * n
        cnt2 = cnt + (cnt & 1)
        u = uniform01(seed + 7919 * (r0 // chunk_rows + 1), cnt2)
        h = cnt2 // 2
        rad = np.sqrt(-2.0 * np.log(u[:h]))
        ang = 2.0 * np.pi * u[h:]
        z = np.empty(cnt2, dtype=np.float64)
        z[0::2] = rad * np.cos(ang)
        z[1::2] = rad * np.sin(ang)
        m[r0:r0 + rows] = (scale * z[:cnt]).reshape(rows, n)
    d = 0.5 + uniform01(seed ^ 0xD1A6, n)
    m[np.arange(n), np.arange(n)] = d
    return m


def write_cuts_bin(path: str, kinds, grads, beta0, beta1) -> None:
    """The cut file host/bench/live_loop.cpp reads: int64 n, int64 k, int32 kinds[k], f64 beta0[k], f64 beta1[k] (NaN: none),
    f64 grads[k][n]."""
    grads = np.ascontiguousarray(grads, dtype=np.float64)
    k, n = grads.shape
    with open(path, "wb") as f:
        np.array([n, k], dtype=np.int64).tofile(f)
        np.ascontiguousarray(kinds, dtype=np.int32)[:k].tofile(f)
        np.ascontiguousarray(beta0, dtype=np.float64)[:k].tofile(f)
        np.ascontiguousarray(beta1, dtype=np.float64)[:k].tofile(f)
        grads.tofile(f)
