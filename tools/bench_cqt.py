"""BASELINE.json configs[2]: batch of 64 x 30 s synthetic polyphonic clips, 84-bin CQT filter bank on the MFMA
units, 1 MI355X.  Prints one JSON line: kernel time (hipEvents), achieved TFLOP/s of issued MFMA work and
fraction of the f32-MFMA peak (157.3 TF, MI355X_MICROARCH.md)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spectrogram_midi_amd import _lib
from tools import signals

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 64
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
base = [signals.polyphonic_clip(secs, seed=100 + i) for i in range(min(n_clips, 8))]
rng = np.random.default_rng(0)
clips = [base[i % len(base)] if i < len(base) else np.roll(base[i % len(base)], int(rng.integers(1, 10000))) for i in range(n_clips)]
h = _lib.Handle()
h.cqt(clips[:2])
h.set_profiling(True)
ms = []
for _ in range(3):
    t0 = time.perf_counter(); out = h.cqt(clips); wall = time.perf_counter() - t0
    ms.append(h.kernel_ms("cqt"))
frames = sum(o.shape[1] for o in out)
# per-tile half supports exactly as csrc/cqt.hip::build_cqt_bank derives them (512-sample granularity)
r = 2.0 ** (1 / 12); alpha = (r * r - 1) / (r * r + 1)
half = []
for T in range(11):
    ilen = (1.0 / alpha) * 44100 / (32.70319566257483 * 2.0 ** (8 * T / 12))
    half.append((int(-np.floor(-ilen / 2)) + 1 + 511) // 512 * 512)
steps = sum(2 * x // 4 for x in half)
tiles = sum(-(-o.shape[1] // 48) for o in out)          # sliding-window kernel: 48 frames (3 column tiles) per workgroup
flops = steps * 2048.0 * tiles * 3                      # one 16x16x4 f32 MFMA = 2048 flop
k = float(np.median(ms))
print(json.dumps({"workload": f"{n_clips} x {secs:g} s polyphonic clips, CQT-84 (C1, 12 bins/octave), hop 512",
                  "frames": frames, "kernel_ms": round(k, 3), "wall_ms_incl_pcie": round(wall * 1e3, 1),
                  "audio_s_per_s_kernel": round(n_clips * secs / (k * 1e-3), 1),
                  "mfma_tflops_issued": round(flops / (k * 1e-3) / 1e12, 2), "peak_f32_mfma_tflops": 157.3,
                  "mfma_frac": round(flops / (k * 1e-3) / 157.3e12, 4)}))
