"""Probe points and constant sets for the LowpassOracle tests (CPU and GPU): together they reach every
return statement of assess_feas (src/oracles/lowpass_oracle.rs:58-133)."""
import math

import numpy as np

from oracle import oracle as O

CONSTANT_SETS = {
    "as_written": O.lowpass_case(False),          # lp_sq > up_sq
    "corrected": O.lowpass_case(True),
    "loose": (0.12, 0.20, 0.5, 1.5, 0.3),
    "very_loose": (0.12, 0.20, 0.0, 4.0, 2.0),
    "negative_passband_allowed": (0.12, 0.20, -1.0, 4.0, 2.0),
}


def lowpass_autocorr(n, cutoff=0.16):
    """autocorrelation r of a windowed-sinc low-pass filter: spectrum @ r = |H(w)|^2 >= 0"""
    k = np.arange(n) - (n - 1) / 2.0
    h = cutoff * np.sinc(cutoff * k) * np.hamming(n)
    h /= h.sum()
    full = np.correlate(h, h, mode="full")
    r = full[n - 1:].copy()
    r[0] += 1e-6  # lift the stopband nulls (|H|^2 = 0 to rounding) clear of the `val < 0` thresholds
    return r


def transition_probe(n):
    """passband and stopband satisfied (very_loose), negative only inside the transition band -> the
    ParallelCut(-val, None) return of :116-121.  Needs n >= 48."""
    r = lowpass_autocorr(n)
    wt = 0.16 * math.pi
    j = np.arange(n)
    b = (1 - j / n) * np.cos(j * wt) / n  # Fejer bump at wt, peak ~0.5, sidelobes ~0.015
    b[0] = 0.5 / n
    e0 = np.zeros(n)
    e0[0] = 1.0
    return r + 0.05 * e0 - 0.75 * b


def negative_x0_probe(n):
    """every band satisfied for `negative_passband_allowed`, x[0] < 0 -> the :126-130 return.  n >= 128."""
    r = lowpass_autocorr(n, cutoff=0.07)
    e0 = np.zeros(n)
    e0[0] = 1.0
    return 0.02 * e0 - 0.5 * r


def probe_points(n, rng, count):
    r = lowpass_autocorr(n)
    e0 = np.zeros(n)
    e0[0] = 1.0
    for it in range(count):
        m = it % 10
        if m == 0:
            yield r.copy()
        elif m == 1:
            yield r - 0.02 * e0
        elif m == 2:
            yield 1.2 * r
        elif m == 3:
            yield e0.copy()
        elif m == 4:
            yield r + 1e-3 * rng.standard_normal(n)
        elif m == 5:
            x = rng.standard_normal(n) * 0.05
            x[0] = -0.2
            yield x
        elif m == 6:
            yield np.zeros(n)
        elif m == 7:
            yield r * (0.5 + rng.random())
        elif m == 8:
            yield transition_probe(n) if n >= 48 else r * 0.9
        else:
            yield negative_x0_probe(n) if n >= 128 else r * 1.1
