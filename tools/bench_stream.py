"""Per-push latency of the streaming path (BASELINE.json configs[4]: 2048-sample pushes = 4 hops, 1 MI355X).
Prints one JSON line with p50 / p99 wall time per push (host call to host return, H2D + kernels + D2H + sync)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spectrogram_midi_amd import _lib
from tools import signals

n_push = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
y = signals.guitar_clip(n_push * 2048 / 44100 + 1.0, seed=1)
h = _lib.Handle()
if os.environ.get("AEGIS_DUMP_MAPS"):          # address -> library map, to symbolise a crash stack offline
    with open(os.environ["AEGIS_DUMP_MAPS"], "w") as f:
        f.write(open("/proc/self/maps").read())
st = h.open_stream(max_seconds=len(y) / 44100 + 1)
lat = []
for i in range(n_push):
    chunk = y[i * 2048:(i + 1) * 2048]
    t0 = time.perf_counter()
    st.push(chunk)
    lat.append(time.perf_counter() - t0)
t0 = time.perf_counter()
out = st.close()
t_close = time.perf_counter() - t0
lat = np.array(lat[50:]) * 1e6
print(json.dumps({"metric": "streaming per-push latency (2048-sample pushes, 4 frames each)", "pushes": len(lat),
                  "p50_us": round(float(np.percentile(lat, 50)), 1), "p99_us": round(float(np.percentile(lat, 99)), 1),
                  "mean_us": round(float(lat.mean()), 1), "realtime_factor": round(2048 / 44100 / (lat.mean() * 1e-6), 1),
                  "close_ms": round(t_close * 1e3, 2), "frames": int(len(out["f0"]))}))
