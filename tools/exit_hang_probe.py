"""Probe for the round-3 record gpurun_out/call53.log: a script that died with an exception while a `Handle` was alive
(module global, 22 050 Hz, profiling on, one analyze_batch) sat until `timeout` killed it.

    python tools/exit_hang_probe.py <variant>       (AEGIS_TRACE_DESTROY=1 prints the teardown step it sits in)

variants: raise  -- global handle, analyze, unhandled exception (the record's case)
          exit   -- global handle, analyze, normal end of the script
          del    -- global handle, analyze, `del h` (destroy in a live interpreter), then the exception
          torch  -- as raise, with torch imported and a CUDA tensor alive as well
"""
import faulthandler
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable()                       # `timeout -s ABRT` then shows the Python frame the process sits in

import numpy as np

from spectrogram_midi_amd import _lib
from tools import signals

variant = sys.argv[1] if len(sys.argv) > 1 else "raise"
if variant == "torch":
    import torch
    keep = torch.zeros(1 << 20, device="cuda")
h = _lib.Handle(sample_rate=22050)
h.set_profiling(True)
y = signals.guitar_clip(180.0, sr=22050, seed=3)     # long enough for time chunks: the CU-masked streams get created
for n in (1,):
    r = h.analyze_batch([y] * n)
    print(variant, n, "clips ok", len(r[0]["f0"]), "frames", flush=True)
if variant == "del":
    del h
    print("deleted", flush=True)
if variant != "exit":
    print(h.no_such_attribute)              # AttributeError (NameError after `del`), as in the record
print("end of script", flush=True)
