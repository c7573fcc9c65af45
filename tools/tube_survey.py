"""The time-split verification on the folder's noisy clips (guitar under a -12 dBFS noise floor: unvoiced throughout, the two
edge bins' unvoiced states tie for the whole clip): verification counters, rails, lock-on statistics, and whether the outputs
equal the sequential run's.

    python tools/tube_survey.py [segment length, default 768]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AEGIS_TIME_SPLIT"] = sys.argv[1] if len(sys.argv) > 1 else "768"
import numpy as np
from spectrogram_midi_amd import _lib
from tools import signals

base = signals.guitar_clip(330.0, 44100, seed=707, noise_dbfs=-12.0)
rng = np.random.default_rng(5)
clips = []
for i in range(8):
    n = int(rng.uniform(60, 180) * 44100)
    o = int(rng.integers(0, len(base) - n))
    clips.append(np.ascontiguousarray(base[o:o + n] * np.float32(rng.uniform(0.5, 1.0))))
h = _lib.Handle()
got = h.analyze_batch(clips)
v = [int(x) for x in h.debug_fetch("split_verify")]
lk = h.debug_fetch("seg_lock")
print("frames per clip (the pass takes them longest first):", sorted((len(c) // 512 + 1 for c in clips), reverse=True))
print("lock-on steps per segment (0: a clip's first segment, -1: never):", lk.tolist())
lk = lk[lk != 0]
names = ("frames", "tubes_opened", "tubes_recorded", "too_wide", "-", "closed_elsewhere", "open_at_exact_run", "too_deep", "max_depth_recorded",
         "oob_in_bound", "last_column_tie", "records_full", "tubes_resolved", "path_changed", "tubes_with_rail", "rail_frames")
print(json.dumps({"clips": len(clips), "frames": int(sum(len(c) // 512 + 1 for c in clips)), "segments": h.param("last_split_segments"),
                  "flagged": h.param("split_flagged_clips"), "never_locked": h.param("split_unlocked_clips"),
                  "verify": dict(zip(names, v)),
                  "lock_on": {"runs": int(len(lk)), "never": int((lk < 0).sum()), "median": float(np.median(lk[lk > 0])) if (lk > 0).any() else None,
                              "p90": float(np.quantile(lk[lk > 0], 0.9)) if (lk > 0).any() else None, "max": int(lk.max()) if len(lk) else None}}))
os.environ["AEGIS_TIME_SPLIT"] = "0"
h2 = _lib.Handle()
ref = h2.analyze_batch(clips)
print("outputs equal the sequential run's:", all(np.array_equal(g[k], r[k], equal_nan=True) for g, r in zip(got, ref) for k in ("f0", "voiced_flag", "voiced_prob")))
