"""Host-side cost of the batched event extraction / SMF rendering (events_native.extract_batch) on frame arrays of the
folder's size (8.36 M frames, 512 clips), with a per-phase profile.  No GPU needed."""
import cProfile
import json
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectrogram_midi_amd import events_native as en   # noqa: E402

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "v1_events_golden.npz"))
keys = ("rake_mask", "f0", "voiced_flag", "voiced_probs", "rms")
base = {k: np.concatenate([g[f"{c}/{k}"] for c in ("notes", "guitar", "scale")]) for k in keys}
rng = np.random.default_rng(0)
lens = (rng.uniform(30, 330, 512) * 86.13).astype(int)
L = len(base["f0"])
cat = {k: np.tile(v, int(lens.sum()) // L + 1)[:lens.sum()] for k, v in base.items()}
off = np.concatenate([[0], np.cumsum(lens)])
freqs = 82.4068892282175 * 2 ** (np.arange(441) / 120)
bins = np.where(cat["voiced_flag"], np.searchsorted(freqs, cat["f0"]).clip(0, 440), -1).astype(np.int16)
res = {"frames": int(lens.sum()), "cpus": os.cpu_count()}
for name, kw in (("log2", {}), ("grid", dict(pitch_bin=bins, freqs=freqs))):
    for rep in range(2):
        t0 = time.perf_counter()
        per, blobs = en.extract_batch(off, cat["rake_mask"], cat["f0"], cat["voiced_flag"], cat["voiced_probs"], cat["rms"], 44100, 512, 0.7,
                                      want_midi=True, **kw)
        dt = time.perf_counter() - t0
    res[name + "_ms"] = round(dt * 1e3, 1)
    res["events"] = sum(len(e) for e in per)
print(json.dumps(res))
pr = cProfile.Profile()
pr.enable()
en.extract_batch(off, cat["rake_mask"], cat["f0"], cat["voiced_flag"], cat["voiced_probs"], cat["rms"], 44100, 512, 0.7, want_midi=True,
                 pitch_bin=bins, freqs=freqs)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(10)
