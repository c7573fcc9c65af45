import sys, os
sys.path.insert(0, os.getcwd())
from spectrogram_midi_amd import _lib
from tools import signals
clips = [signals.polyphonic_clip(30.0, seed=100 + i % 4) for i in range(64)]
h = _lib.Handle(); h.cqt(clips[:2]); h.cqt(clips)
v = h.debug_fetch("cqt_cycles")
import numpy as np
r = 2.0 ** (1 / 12); alpha = (r * r - 1) / (r * r + 1)
half = [(int(-np.floor(-((1.0 / alpha) * 44100 / (32.70319566257483 * 2.0 ** (8 * T / 12))) / 2)) + 1 + 511) // 512 * 512 for T in range(11)]
npass = 2 * half[0] // 256
passes = {}                                                        # 256-tap passes per active-tile count (CQT-84), as build_cqt_bank lays them out
for q in range(npass):
    na = sum(1 for h_ in half if (half[0] - h_) // 256 <= min(q, npass - 1 - q))
    passes[na] = passes.get(na, 0) + 1
print("prologue", int(v[0]), "barrier", int(v[12]), "epilogue", int(v[13]), "total", int(v[14]))
for k in range(1, 12):
    if v[k]:
        n = passes.get(k, 0)
        print(f"na={k:2d}: {int(v[k]):8d} cycles" + (f" = {v[k] / n:8.0f}/pass = {v[k] / n / 16:6.0f}/k-step = {v[k] / n / 16 / (3 * k):5.1f}/MFMA" if n else ""))
