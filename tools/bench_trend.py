"""Runs every aegis_trend op once over a batch of 64 pitch tracks of 15 504 frames (a 64 x 180 s batch) and over one
single track (the v2 engine's one-clip-per-request shape), for `rocprofv3 --kernel-trace --stats -- python3
tools/bench_trend.py`.  Prints wall time per op (host call to host return, H2D + kernels + D2H)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.signal import savgol_coeffs
from spectrogram_midi_amd import _lib

rng = np.random.default_rng(0)
F = 15504


def track():
    midi = np.repeat(rng.integers(40, 84, F // 40 + 1), 40)[:F] + rng.normal(0, 0.05, F)
    f0 = 440.0 * 2 ** ((midi - 69) / 12)
    f0[rng.random(F) < 0.25] = np.nan
    return f0


h = _lib.Handle()
coef = savgol_coeffs(11, 3)[::-1]
ops = [("sma", _lib.TREND_SMA, [10], 1, np.float64), ("ema", _lib.TREND_EMA, [12], 1, np.float64),
       ("bollinger", _lib.TREND_BOLLINGER, [20, 2.0], 3, np.float64), ("articulation", _lib.TREND_ARTICULATION, [20, 2.0], 1, np.int8),
       ("macd", _lib.TREND_MACD, [12, 26, 9], 3, np.float64), ("slides", _lib.TREND_SLIDES, [0.5], 1, np.int8),
       ("rsi", _lib.TREND_RSI, [14], 1, np.float64), ("savgol", _lib.TREND_SAVGOL, [11, 1, *coef], 1, np.float64),
       ("kalman", _lib.TREND_KALMAN, [1e-5, 1e-1], 1, np.float64), ("holt", _lib.TREND_HOLT, [0.3, 0.1], 1, np.float64)]
out = {}
for label, series in (("batch64", [track() for _ in range(64)]), ("single", [track()])):
    for name, op, par, n_out, dt in ops:
        h.trend(op, series, par, n_out=n_out, out_dtype=dt)
        t0 = time.perf_counter()
        h.trend(op, series, par, n_out=n_out, out_dtype=dt)
        out[f"{label}/{name}_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
print(json.dumps(out))
