"""End-to-end through the reference-shaped surface: AegisEngine.analyze_arrays (host NumPy in, raw_data dicts out)
and extract_events (+ SMF bytes) on the bench workload."""
import io, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from spectrogram_midi_amd.engine import AegisEngine

clips = bench.make_clips(64, 180.0, seed0=1)
eng = AegisEngine()
eng.analyze_arrays(clips)          # warm: workspaces and staging buffers at their final size
ta, te, tm = [], [], []
for _ in range(5):
    t0 = time.perf_counter(); raws = eng.analyze_arrays(clips); t1 = time.perf_counter()
    evs = [eng.extract_events(r, None) for r in raws]; t2 = time.perf_counter()
    for r in raws:
        eng.extract_events(r, io.BytesIO())
    t3 = time.perf_counter()
    ta.append(t1 - t0); te.append(t2 - t1); tm.append(t3 - t2)
a, e, m = (float(np.median(x)) for x in (ta, te, tm))
tb = []
for _ in range(5):
    t0 = time.perf_counter(); eng.audio_to_midi_batch(clips); tb.append(time.perf_counter() - t0)
b = float(np.median(tb))
print(json.dumps({"workload": "64 x 180 s through AegisEngine (host arrays in, dicts / events / SMF bytes out)",
                  "audio_to_midi_batch_s": round(b, 4), "audio_s_per_s_batch_to_midi": round(64 * 180 / b, 1),
                  "analyze_arrays_s": round(a, 4), "extract_events_s": round(e, 4), "extract_events_with_smf_s": round(m, 4),
                  "events_per_clip": int(np.mean([len(x) for x in evs])),
                  "audio_s_per_s_analyze": round(64 * 180 / a, 1), "audio_s_per_s_to_midi": round(64 * 180 / (a + m), 1),
                  "analyze_arrays_all_s": [round(x, 4) for x in ta], "audio_to_midi_batch_all_s": [round(x, 4) for x in tb]}))
