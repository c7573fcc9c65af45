"""GPU-box diagnostic: stage-by-stage comparison of one clip against the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyin as opyin
from spectrogram_midi_amd import _lib
from tools import signals

rng = np.random.default_rng(0)
dur = float(rng.uniform(2.0, 30.0))
y = signals.polyphonic_clip(dur, seed=100)
h = _lib.Handle()
r = h.analyze_batch([y])[0]
yin_g = h.debug_fetch("yin").reshape(-1, h.param("yin_stride"))[:, :495]
lo_g = h.debug_fetch("logobs").reshape(-1, h.param("obs_stride"))[:, :441]
lu_g = h.debug_fetch("logunv")
st_g = h.debug_fetch("states")
f0, vf, vp, it = opyin.pyin(y, return_intermediates=True)
print("frames", len(f0), "voiced mismatches", np.flatnonzero(r["voiced_flag"] != vf))
yin_o = it["yin"].T
rel = np.abs(yin_g - yin_o) / np.maximum(np.abs(yin_o), 1e-12)
print("yin max rel", rel.max(), "at", np.unravel_index(rel.argmax(), rel.shape))
obs = it["obs"]
lo_o = np.log(obs[:441].T + opyin.TINY)
sup = (lo_g > -700) != (lo_o > -700)
print("support diffs", np.argwhere(sup)[:20])
d = np.abs(lo_g - lo_o); d[sup] = 0
print("logobs max abs (common support)", d.max(), np.unravel_index(d.argmax(), d.shape))
lu_o = np.log(obs[441] + opyin.TINY)
print("logunv max abs", np.abs(lu_g - lu_o).max(), np.argmax(np.abs(lu_g - lu_o)), "exp diff", np.abs(np.exp(lu_g) - obs[441]).max())
sd = np.flatnonzero(st_g != it["states"])
print("state diffs", len(sd), sd[:40])
# Viterbi on GPU observations with the oracle decoder: isolates the decoder
obs2 = np.zeros_like(obs); obs2[:441] = np.exp(lo_g.T) - 0.0; obs2[:441][lo_g.T < -700] = 0.0; obs2[441:] = np.exp(lu_g)[None, :]
p = it["params"]
trans = opyin.transition_matrix(p)
S = 882
lp = np.vstack([lo_g.T, np.repeat(lu_g[None, :], 441, 0)]).T
st2 = opyin.viterbi_states(lp, np.log(trans + opyin.TINY), np.log(np.ones(S) / S + opyin.TINY))
print("oracle-decoder on GPU log-obs vs GPU states: diffs", np.count_nonzero(st2 != st_g), "vs oracle states", np.count_nonzero(st2 != it["states"]))
dd = np.flatnonzero(st2 != st_g)
print(dd[:30])
if len(dd):
    t = dd[0]
    print("t", t, "gpu", st_g[t-2:t+3], "cpu-dec", st2[t-2:t+3], "oracle", it["states"][t-2:t+3])
