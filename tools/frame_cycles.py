"""Section cycle counters of frame_yin_kernel (workgroup 1000, threads 0 and 64, summed over its 8 frame pairs); needs a
-DAEGIS_ABLATE=128 build:  make -C spectrogram-midi_amd/csrc EXTRA=-DAEGIS_ABLATE=128 OBJDIR=_obj128 OUT=../../_ablate/lib_ab128.so
AEGIS_HIP_LIB=_ablate/lib_ab128.so python tools/frame_cycles.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spectrogram_midi_amd import _lib
from tools import signals
y = signals.guitar_clip(180.0, seed=1)
clips = [np.roll(y, 1000 * i) for i in range(64)]
h = _lib.Handle()
h.analyze_batch(clips)
v = h.debug_fetch("frame_cycles")
names = ["energy prologue", "wait samples", "rms + fwd fft", "separate + power", "mel", "inverse fft", "difference store",
         "CMND: rows back", "CMND: cumsum walk", "CMND: quotients"]
for base, who in ((0, "thread 0 (wave 0)"), (12, "thread 64 (wave 1)")):
    tot = int(v[base:base + 10].sum())
    print(who, "total", tot, "clock64 ticks" )
    for n, c in zip(names, v[base:base + 10]):
        print(f"  {n:18s} {int(c):9d}  {100.0 * c / max(tot, 1):5.1f} %")
