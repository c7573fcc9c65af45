"""Where the host-buffer entry's time goes on the 512-clip folder: the C call alone on outputs whose
pages exist, the Python binding around it, the event extraction and the MIDI rendering, (round 4: page-locked clips, a gather kernel over mapped host memory, two copy streams, larger feed chunks
and sub-batches with the host work on a helper thread were all measured with this script and bought nothing: DESIGN.md section 7)."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from spectrogram_midi_amd import _lib, events_native
from spectrogram_midi_amd.engine import AegisEngine

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 512
durations = bench.folder_durations(n_clips)
clips = bench.make_folder_clips(range(n_clips), durations)
audio = sum(len(c) for c in clips) / 44100
h = _lib.Handle()
lib = h.lib
F = sum(1 + len(c) // 512 for c in clips)


def c_call(cl):
    n = len(cl)
    ptrs = (C.c_void_p * n)(*[c.ctypes.data for c in cl])
    lens = (C.c_int64 * n)(*[len(c) for c in cl])
    bufs = {"f0": np.zeros(F, np.float64), "voiced_flag": np.zeros(F, np.uint8), "voiced_prob": np.zeros(F, np.float64),
            "rms": np.zeros(F, np.float32), "rake_mask": np.zeros(F, np.uint8), "pitch_bin": np.zeros(F, np.int16)}
    out = _lib.Outputs()
    for k, v in bufs.items():
        setattr(out, k, v.ctypes.data)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        rc = lib.aegis_analyze_batch(h._h, ptrs, lens, n, 0.6, _lib.STAGE_ALL | _lib.OPT_F0_ZERO, C.byref(out))
        ts.append(time.perf_counter() - t0)
        assert rc == 0
    return min(ts[1:]) * 1e3


def binding(cl):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        h.analyze_batch(cl, want_sdb=False, f0_zero=True, views=True, concatenated=True)
        ts.append(time.perf_counter() - t0)
    return min(ts[1:]) * 1e3


res = {"clips": n_clips, "audio_s": round(audio, 1)}
res["pageable"] = {"c_call_ms": round(c_call(clips), 1), "binding_ms": round(binding(clips), 1)}
eng = AegisEngine()
eng._handle = h
ts = []
for _ in range(2):
    t0 = time.perf_counter()
    eng.audio_to_midi_batch(clips)
    ts.append(time.perf_counter() - t0)
res["audio_to_midi_batch_ms"] = round(min(ts) * 1e3, 1)
r, bufs, off = h.analyze_batch(clips, want_sdb=False, f0_zero=True, views=True, concatenated=True)
t0 = time.perf_counter()
events_native.extract_batch(off, bufs["rake_mask"], bufs["f0"], bufs["voiced_flag"], bufs["voiced_prob"], bufs["rms"], 44100, 512, 0.70,
                            want_midi=True, pitch_bin=bufs["pitch_bin"], freqs=h.table("freqs"))
res["events_and_smf_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
eng._handle = None
print(json.dumps(res))
