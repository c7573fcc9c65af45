"""Probe for the round-3 record gpurun_out/call53.log: a script that died with an exception while a `Handle` was alive
(module global, 22 050 Hz, profiling on, one analyze_batch of one 180 s clip) sat until `timeout` killed it.

    python tools/exit_hang_probe.py <variant> [sr] [clips] [seconds] [profiling 0|1] [alarm seconds, 0 = none] [host|device]

(AEGIS_TRACE_DESTROY=1 prints the teardown step it sits in)
variants: raise  -- global handle, analyze, unhandled exception (the record's case)
          exit   -- global handle, analyze, normal end of the script
          del    -- global handle, analyze, `del h` (destroy in a live interpreter), then the exception
          close  -- explicit h.close(), then a normal end
"""
import faulthandler
import os
import signal
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable()                       # `timeout -s ABRT` then shows the Python frame the process sits in

import numpy as np

from spectrogram_midi_amd import _lib
from tools import signals

variant = sys.argv[1] if len(sys.argv) > 1 else "raise"
sr = int(sys.argv[2]) if len(sys.argv) > 2 else 22050
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1
seconds = float(sys.argv[4]) if len(sys.argv) > 4 else 180.0
prof = (sys.argv[5] != "0") if len(sys.argv) > 5 else True
if len(sys.argv) > 6 and int(sys.argv[6]) > 0:
    signal.alarm(int(sys.argv[6]))          # under a debugger: SIGALRM stops the process where it sits
h = _lib.Handle(sample_rate=sr)
h.set_profiling(prof)
y = signals.guitar_clip(seconds, sr=sr, seed=3)
entry = sys.argv[7] if len(sys.argv) > 7 else "host"
if entry == "device":
    import torch
    F = n * (1 + len(y) // 512)
    d_pcm = torch.from_numpy(np.concatenate([y] * n)).cuda()
    outs = {"f0": torch.empty(F, dtype=torch.float64, device="cuda"), "voiced_flag": torch.empty(F, dtype=torch.uint8, device="cuda"),
            "voiced_prob": torch.empty(F, dtype=torch.float64, device="cuda"), "rms": torch.empty(F, dtype=torch.float32, device="cuda"),
            "rake_mask": torch.empty(F, dtype=torch.uint8, device="cuda")}
    h.analyze_batch_device(d_pcm.data_ptr(), np.arange(n + 1, dtype=np.int64) * len(y), {k: v.data_ptr() for k, v in outs.items()}, sync=True)
    r = [{"f0": outs["f0"].cpu().numpy()[:F // n]}]
else:
    r = h.analyze_batch([y] * n)
print(variant, entry, sr, n, seconds, prof, "ok", len(r[0]["f0"]), "frames; passes", h.param("last_passes"), "chunks", h.param("last_chunks"),
      "balanced", h.param("last_balanced"), "persistent", h.param("last_persistent"), flush=True)
if variant == "del":
    del h
    print("deleted", flush=True)
if variant == "close":
    h.close()
    print("closed", flush=True)
elif variant != "exit":
    print(h.no_such_attribute)              # AttributeError (NameError after `del`), as in the record
print("end of script", flush=True)
