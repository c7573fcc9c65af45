"""Analyze path at the v2 engine's default rate (22 050 Hz, hop 512: transition width 101, viterbi_band_kernel<50>):
64 x 180 s through the host-buffer entry; prints wall time per batch and the hipEvent kernel times."""
import sys, os, time
import numpy as np, torch
torch.zeros(1, device="cuda")          # torch first: it must see the GPU before the library has opened it
sys.path.insert(0, os.getcwd())
from spectrogram_midi_amd import _lib
from tools import signals
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

# the same batch through the device-pointer entry (PCM and outputs resident in HBM: CU-partitioned, balanced schedule)
dev = torch.device("cuda", 0)
n = np.array([len(c) for c in clips], np.int64)
off = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
F = int(sum(1 + len(c) // 512 for c in clips))
d_pcm = torch.from_numpy(np.concatenate(clips)).to(dev)
outs = {"f0": torch.empty(F, dtype=torch.float64, device=dev), "voiced_flag": torch.empty(F, dtype=torch.uint8, device=dev),
        "voiced_prob": torch.empty(F, dtype=torch.float64, device=dev), "rms": torch.empty(F, dtype=torch.float32, device=dev),
        "rake_mask": torch.empty(F, dtype=torch.uint8, device=dev)}
ptrs = {k: v.data_ptr() for k, v in outs.items()}
h.analyze_batch_device(d_pcm.data_ptr(), off, ptrs, sync=True)
ts = []
for _ in range(4):
    t0 = time.perf_counter(); h.analyze_batch_device(d_pcm.data_ptr(), off, ptrs, sync=True); ts.append(time.perf_counter() - t0)
print("22.05k device entry: s/step", min(ts), "audio-s/s", 64 * 180 / min(ts), {k: round(h.kernel_ms(k), 1) for k in ("frame", "pyin_obs", "viterbi", "finalize")},
      "viterbi launches", h.kernel_launches("viterbi"))
