"""`_pyin_worker((chunk, sr, hop_length)) -> (f0, voiced_flag, voiced_prob)`: the picklable
Turbo-Mode worker of the reference (/root/reference/aegis_engine_core/worker.py:3-15), served
by the GPU.  Handles are cached per (sr, hop) in the calling process."""
import numpy as np

from . import _lib
from .convert import note_to_hz

_handles = {}
PYIN_INIT = "unvoiced"      # initial distribution of the HMM (see _lib.Handle); read when a handle is first created


def _pyin_worker(args):
    chunk, sr, hop_length = args
    key = (int(sr), int(hop_length), PYIN_INIT)
    if key not in _handles:
        _handles[key] = _lib.Handle(sample_rate=key[0], hop_length=key[1], fmin=note_to_hz("E2"),
                                    fmax=note_to_hz("C6"), pyin_init=PYIN_INIT)
    r = _handles[key].analyze_batch([np.asarray(chunk, np.float32)], stages=_lib.STAGE_PYIN)[0]
    return r["f0"], r["voiced_flag"], r["voiced_prob"]
