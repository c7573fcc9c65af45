import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import cqt as ocqt
from spectrogram_midi_amd import _lib, signals
y = signals.polyphonic_clip(2.5, seed=101)
for hop in (1024, 512):
    h = _lib.Handle(hop_length=hop, scipy_tables=False)
    got = h.cqt([y])[0]
    ref = np.abs(ocqt.cqt(y, hop_length=hop))
    err = np.abs(got - ref)
    print(hop, "max err", err.max(), "ref max", ref.max())
    print(" per-tile max err", [float(err[8*T:8*T+8].max().round(4)) for T in range(11)])
    print(" per-tile ref max", [float(ref[8*T:8*T+8].max().round(4)) for T in range(11)])
    h.close()
