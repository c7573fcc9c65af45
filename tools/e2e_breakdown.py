"""Where AegisEngine.audio_to_midi_batch spends its time on the 512-clip folder (bench.py's engine_e2e record):
the analysis call, then the parts of events_native.extract_batch (dB track + gate, C++ run detection, SMF rendering,
the reference's list-of-dicts schema).  python tools/e2e_breakdown.py [n_clips]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from spectrogram_midi_amd import events_native as en
from spectrogram_midi_amd.engine import AegisEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 512
dur = bench.folder_durations(n)
clips = bench.make_folder_clips(list(range(n)), dur)
audio = float(sum(len(c) for c in clips)) / bench.SR
eng = AegisEngine()
T = {}


def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        T[name] = T.get(name, 0.0) + (time.perf_counter() - t) * 1e3
        return r
    return w


lib = en._bind()


class Lib:
    def __getattr__(self, k):
        v = getattr(lib, k)
        return timed(k, v) if k in ("aegis_extract_events", "aegis_render_smf") else v


proxy = Lib()
en._bind = lambda: proxy
en.batch_rms_db = timed("rms_db_and_gate", en.batch_rms_db)
en_extract = en.extract_batch
en.extract_batch = timed("extract_batch", en_extract)
eng.analyze_arrays = timed("analyze_arrays", eng.analyze_arrays)
rows = []
for it in range(4):
    T.clear()
    t0 = time.perf_counter()
    raws, evs, blobs = eng.audio_to_midi_batch(clips)
    T["audio_to_midi_batch"] = (time.perf_counter() - t0) * 1e3
    T["events"] = sum(len(e) for e in evs)
    del raws, evs, blobs
    rows.append({k: round(v, 2) for k, v in T.items()})
if "--profile" in sys.argv:
    import cProfile, pstats
    raws, bufs, off, live = eng.analyze_arrays(clips, _concatenated=True)
    pr = cProfile.Profile()
    pr.enable()
    en_extract(off, bufs["rake_mask"], bufs["f0"], bufs["voiced_flag"], bufs["voiced_prob"], bufs["rms"], eng.sr, eng.hop_length, 0.70,
               want_midi=True, pitch_bin=bufs["pitch_bin"], freqs=eng.handle.table("freqs"))
    pr.disable()
    pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(14)
best = min(rows[1:], key=lambda r: r["audio_to_midi_batch"])
print(json.dumps({"clips": n, "audio_s": round(audio, 1), "host_workers": en._host_workers(), "runs": rows,
                  "audio_s_per_s": round(audio / (best["audio_to_midi_batch"] / 1e3), 1)}, indent=1))
eng.close()
