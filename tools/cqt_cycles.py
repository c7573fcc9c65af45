import sys, os
sys.path.insert(0, os.getcwd())
from spectrogram_midi_amd import _lib
from tools import signals
clips = [signals.polyphonic_clip(30.0, seed=100 + i % 4) for i in range(64)]
h = _lib.Handle(); h.cqt(clips[:2]); h.cqt(clips)
v = h.debug_fetch("cqt_cycles")
passes = {1: 32, 2: 22, 3: 14, 4: 8, 5: 4, 6: 4, 7: 4, 11: 4}     # 256-tap passes per active-tile count (CQT-84)
print("prologue", int(v[0]), "barrier", int(v[12]), "epilogue", int(v[13]), "total", int(v[14]))
for k in range(1, 12):
    if v[k]:
        n = passes.get(k, 0)
        print(f"na={k:2d}: {int(v[k]):8d} cycles" + (f" = {v[k] / n:8.0f}/pass = {v[k] / n / 16:6.0f}/k-step = {v[k] / n / 16 / (3 * k):5.1f}/MFMA" if n else ""))
