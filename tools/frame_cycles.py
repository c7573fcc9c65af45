"""Section cycle counters of frame_fft_kernel (workgroup 1000, thread 0, summed over its 4 frame pairs), needs a -DAEGIS_ABLATE=128 build:
AEGIS_HIP_LIB=_ablate/lib_ab128.so python tools/frame_cycles.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectrogram_midi_amd import _lib, signals
clips = [signals.guitar_clip(60.0, seed=1 + i % 4) for i in range(16)]
h = _lib.Handle()
h.analyze_batch(clips)
v = h.debug_fetch("frame_cycles")
names = ["load", "rms", "fwd fft", "separate+power", "mel", "inverse fft", "acf store"]
for n, x in zip(names, v):
    print(f"{n:16s} {int(x):8d}")
print("total", int(v[:7].sum()))
