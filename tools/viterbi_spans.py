"""Span statistics of the unvoiced-source arg-max in viterbi_band_kernel (first 8 workgroups), needs a
-DAEGIS_ABLATE=512 build:  AEGIS_HIP_LIB=_ablate/lib_ab512.so python tools/viterbi_spans.py
The interior row of the transition table is log(c * triangle + tiny), concave in the offset, so the lowest-index
arg-max source r(b') is non-decreasing in the target bin b'; the histograms say how far apart the arg-maxes of a wave's
(and of 8 neighbouring) targets are -- what a two-level search would have to cover."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectrogram_midi_amd import _lib   # noqa: E402
from tools import signals   # noqa: E402

sets = {"guitar": [signals.guitar_clip(60.0, seed=1 + i) for i in range(8)],
        "polyphonic": [signals.polyphonic_clip(60.0, seed=700 + i) for i in range(8)],
        "noisy": [signals.guitar_clip(60.0, seed=40 + i, noise_dbfs=-12.0) for i in range(8)]}
h = _lib.Handle()
out = {}
for name, clips in sets.items():
    h.debug_fetch("viterbi_spans")
    h.analyze_batch(clips, stages=_lib.STAGE_PYIN)
    v = h.debug_fetch("viterbi_spans")
    wave, g8, off = v[:128], v[128:192], v[200:264]
    def q(hist, p):
        c = np.cumsum(hist) / max(1, hist.sum())
        return int(np.searchsorted(c, p))
    out[name] = {"wave_steps": int(v[193]), "groups_of_8": int(v[194]), "decreasing_neighbour_pairs": int(v[192]),
                 "wave_span": {"mean": round(float((np.arange(128) * wave).sum() / max(1, wave.sum())), 2),
                               "p50": q(wave, .5), "p90": q(wave, .9), "p99": q(wave, .99), "max": int(np.nonzero(wave)[0].max())},
                 "group8_span": {"mean": round(float((np.arange(64) * g8).sum() / max(1, g8.sum())), 2),
                                 "p50": q(g8, .5), "p90": q(g8, .9), "p99": q(g8, .99), "max": int(np.nonzero(g8)[0].max()),
                                 "hist": [int(x) for x in g8[:20]]},
                 "abs_offset": {"mean": round(float((np.arange(64) * off).sum() / max(1, off.sum())), 2),
                                "hist": [int(x) for x in off[:26]]}}
print(json.dumps(out, indent=1))
