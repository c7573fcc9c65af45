"""Per-wave section cycle counters of viterbi_band_kernel (workgroup 0) on the bench workload (64 x 180 s), once with the
time-chunked pipeline (frame stage running beside the Viterbi) and once with a single chunk (Viterbi alone);
needs a -DAEGIS_ABLATE=64 build:  AEGIS_HIP_LIB=_ablate/lib_ab64.so python tools/viterbi_cycles_load.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from spectrogram_midi_amd import _lib
clips = bench.make_clips(64, 180.0, seed0=1)
names = ["unvoiced chain+edges", "voiced sources", "combine+stores", "end_of_step", "chunk maps", "loop top"]
for label, chunk in (("pipelined (frame stage beside it)", None), ("single chunk (alone)", "16384")):
    if chunk:
        os.environ["AEGIS_TIME_CHUNK"] = chunk
    h = _lib.Handle(max_frames_per_pass=1 << 21)
    h.analyze_batch(clips[:2])
    h.debug_fetch("viterbi_cycles")            # reset
    h.analyze_batch(clips, want_sdb=False)
    v = h.debug_fetch("viterbi_cycles").reshape(16, 8)
    print(label)
    print("wave  steps " + "  ".join(f"{n:>20s}" for n in names) + "   total/step")
    for w in range(16):
        n = v[w, 7]
        if n:
            print(f"{w:4d} {n:6d} " + "  ".join(f"{v[w, k] / n:20.0f}" for k in range(6)) + f"   {v[w, :6].sum() / n:8.0f}")
    h.close()
