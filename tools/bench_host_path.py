"""PCIe-inclusive throughput of the host-buffer entry (aegis_analyze_batch: pageable NumPy arrays in, NumPy arrays
out) on the bench workload.  For DESIGN.md only -- never the `value` of bench.py."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from spectrogram_midi_amd import _lib

clips = bench.make_clips(64, 180.0, seed0=1)
h = _lib.Handle()
h.analyze_batch(clips, want_sdb=False)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); h.analyze_batch(clips, want_sdb=False); ts.append(time.perf_counter() - t0)
t = float(np.median(ts))
print(json.dumps({"workload": "64 x 180 s, host NumPy buffers in and out (H2D 2.03 GB + D2H 22 MB + Python unpacking)",
                  "s_per_step": round(t, 4), "audio_s_per_s": round(64 * 180 / t, 1)}))
