"""Per-wave section cycle counters of viterbi_band_kernel (workgroup 0), needs a -DAEGIS_ABLATE=64 build:
AEGIS_HIP_LIB=_ablate/lib_ab64.so python tools/viterbi_cycles.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectrogram_midi_amd import _lib
from tools import signals
clips = [signals.guitar_clip(60.0, seed=1 + i) for i in range(4)]
h = _lib.Handle()
h.analyze_batch(clips[:1])
h.debug_fetch("viterbi_cycles")            # reset
h.analyze_batch(clips)
v = h.debug_fetch("viterbi_cycles").reshape(16, 8)
names = ["unvoiced chain+edges", "voiced sources", "combine+stores", "end_of_step", "chunk maps", "loop top"]
print("wave  steps " + "  ".join(f"{n:>20s}" for n in names) + "   total/step")
for w in range(16):
    n = v[w, 7]
    if n:
        print(f"{w:4d} {n:6d} " + "  ".join(f"{v[w, k] / n:20.0f}" for k in range(6)) + f"   {v[w, :6].sum() / n:8.0f}")
