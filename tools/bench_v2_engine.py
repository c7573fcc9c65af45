"""The v2 engine (AegisFinancialEngine, 22 050 Hz) on 64 x 180 s: the batched facade against clip-by-clip calls."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spectrogram_midi_amd.engine_financial import AegisFinancialEngine
from tools import signals

sr = 22050
base = [signals.guitar_clip(180.0, seed=1 + i, sr=sr) for i in range(8)]
rng = np.random.default_rng(0)
clips = [base[i % 8] if i < 8 else (np.roll(base[i % 8], int(rng.integers(1, len(base[0])))) * np.float32(rng.uniform(0.5, 1))).astype(np.float32)
         for i in range(64)]
eng = AegisFinancialEngine()
eng.analyze_arrays(clips[:4])
tb = []
for _ in range(3):
    t0 = time.perf_counter(); evs = eng.analyze_arrays(clips); tb.append(time.perf_counter() - t0)
t0 = time.perf_counter()
one = [eng.analyze_array(c) for c in clips[:8]]
t1 = (time.perf_counter() - t0) / 8
ts = []
for _ in range(5):
    t0 = time.perf_counter(); eng.analyze_array(clips[0]); ts.append(time.perf_counter() - t0)
b = float(np.median(tb))
print(json.dumps({"workload": "64 x 180 s @ 22 050 Hz through AegisFinancialEngine (audio -> v2 note events)",
                  "batched_s": round(b, 4), "audio_s_per_s_batched": round(64 * 180 / b, 1),
                  "clip_by_clip_s_per_clip": round(t1, 4), "audio_s_per_s_clip_by_clip": round(180 / t1, 1),
                  "single_clip_ms": round(float(np.median(ts)) * 1e3, 2), "events": int(sum(len(e) for e in evs))}))
if len(sys.argv) > 1 and sys.argv[1] == "--profile":
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); eng.analyze_arrays(clips); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
    pr = cProfile.Profile(); pr.enable(); eng.analyze_array(clips[0]); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(10)
