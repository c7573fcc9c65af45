"""Oracle: rake (broadband burst) mask (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates `/root/reference/aegis_engine_core/vision.py:3-38`
(`detect_rake_patterns(S_dB, hop_length, sr, broadband_threshold_ratio)`).
PINNED: tests/golden/rake_*.npz were produced by importing the reference module
itself in the build container (tests/golden/make_rake_golden.py).
"""
import numpy as np


def broadband_columns(S_dB, ratio):
    """vision.py:11-21 -- column t is a candidate when its peak is >= -60 dB and
    more than `ratio` of the bands lie within 20 dB of that peak."""
    S_dB = np.asarray(S_dB)
    n_bands = S_dB.shape[0]
    peak = S_dB.max(axis=0)
    active = (S_dB > (peak - 20)[None, :]).sum(axis=0)
    return (~(peak < -60)) & ((active / n_bands) > ratio)


def run_length_window(hop_length, sr):
    """vision.py:23-25 -- (min_frames, max_frames) for 10..30 ms."""
    ms_per_frame = (hop_length / sr) * 1000
    return int(10 / ms_per_frame), int(30 / ms_per_frame)


def keep_short_runs(flags, min_frames, max_frames):
    """vision.py:27-36 -- keep closed runs with min<=len<=max; a run still open
    at the end of the array is dropped."""
    flags = np.asarray(flags, dtype=bool)
    out = np.zeros_like(flags)
    edges = np.diff(np.concatenate(([0], flags.view(np.int8), [0])))
    starts, stops = np.nonzero(edges == 1)[0], np.nonzero(edges == -1)[0]
    for s, e in zip(starts, stops):
        if e < len(flags) and min_frames <= e - s <= max_frames:
            out[s:e] = True
    return out


def detect_rake_patterns(S_dB, hop_length, sr, broadband_threshold_ratio):
    lo, hi = run_length_window(hop_length, sr)
    return keep_short_runs(broadband_columns(S_dB, broadband_threshold_ratio), lo, hi)
