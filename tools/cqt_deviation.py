"""How far is librosa's constant-Q transform (octave-recursive, oracle/cqt_recursive.py) from the direct transform the
GPU and oracle/cqt.py compute?  CPU only.  Prints max / mean deviations of |C|, of the chroma and of the auto-matcher's
similarity score (auto_matcher.py:52-83) on the test clips; DESIGN.md section 3.8 quotes them."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import chroma as oc, cqt as od, cqt_recursive as orc, dsp   # noqa: E402
from tools import signals   # noqa: E402

clips = {"guitar_fixture": signals.guitar_test_track(), "polyphonic": signals.polyphonic_clip(3.0, seed=100),
         "notes": signals.guitar_clip(3.0, seed=11), "scale": signals.c_major_scale(44100)[:3 * 44100]}
out = {}
for name, y in clips.items():
    a, b = np.abs(od.cqt(y)), np.abs(orc.cqt(y))
    n = min(a.shape[1], b.shape[1])
    a, b = a[:, :n], b[:, :n]
    ca, cb = oc.chroma_cqt(y), orc.chroma_cqt(y)
    out[name] = {"cqt84_max_abs_dev_over_max": float(np.abs(a - b).max() / a.max()),
                 "cqt84_rel_l2": float(np.linalg.norm(a - b) / np.linalg.norm(a)),
                 "chroma_max_abs_dev": float(np.abs(ca - cb).max()), "chroma_mean_abs_dev": float(np.abs(ca - cb).mean()),
                 "chroma_cosine_direct_vs_recursive": oc.cosine(ca, cb)}
# the similarity score between pairs of clips: direct chroma vs recursive chroma in the 0.6-weighted term
names = list(clips)
pairs = {}
for i in range(len(names)):
    for j in range(i + 1, len(names)):
        ya, yb = clips[names[i]], clips[names[j]]
        n = min(len(ya), len(yb))
        ya, yb = ya[:n], yb[:n]
        mel = oc.cosine(dsp.melspectrogram(ya), dsp.melspectrogram(yb))
        sd = 0.4 * mel + 0.6 * oc.cosine(oc.chroma_cqt(ya), oc.chroma_cqt(yb))
        sr_ = 0.4 * mel + 0.6 * oc.cosine(orc.chroma_cqt(ya), orc.chroma_cqt(yb))
        pairs[f"{names[i]}~{names[j]}"] = {"score_direct": round(sd, 6), "score_recursive": round(sr_, 6), "abs_diff": round(abs(sd - sr_), 6)}
out["similarity_pairs"] = pairs
out["max_score_diff"] = max(p["abs_diff"] for p in pairs.values())
print(json.dumps(out, indent=1))
