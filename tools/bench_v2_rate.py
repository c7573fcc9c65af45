"""Analyze path at the v2 engine's default rate (22 050 Hz, hop 512: transition width 101, viterbi_band_kernel<50>):
64 x 180 s through the host-buffer entry; prints wall time per batch and the hipEvent kernel times."""
import sys, os, time
sys.path.insert(0, os.getcwd())
from spectrogram_midi_amd import _lib, signals
sr = 22050
clips = [signals.guitar_clip(180.0, seed=1 + i % 8, sr=sr) if "sr" in signals.guitar_clip.__code__.co_varnames else None for i in range(64)]
if clips[0] is None:
    base = [signals.guitar_clip(180.0, seed=1 + i)[::2].copy() for i in range(8)]
    clips = [base[i % 8] for i in range(64)]
h = _lib.Handle(sample_rate=sr)
h.analyze_batch(clips[:2], want_sdb=False)
h.set_profiling(True)
ts = []
for _ in range(2):
    t0 = time.perf_counter(); h.analyze_batch(clips, want_sdb=False); ts.append(time.perf_counter() - t0)
print("22.05k: s/step", min(ts), "audio-s/s", 64 * 180 / min(ts), {k: round(h.kernel_ms(k), 1) for k in ("frame", "pyin_obs", "viterbi", "finalize")})
