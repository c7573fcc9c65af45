"""Microseconds per Viterbi step by material (sequential band kernel alone on the chip, 16 clips x 120 s of one kind)."""
import json, os, sys
for k, v in (("AEGIS_TIME_CHUNK", "100000000"), ("AEGIS_CU_SPLIT", "0"), ("AEGIS_TIME_SPLIT", "0"), ("AEGIS_DENSE", "0"), ("AEGIS_PROPORTIONAL_CHUNKS", "0")):
    os.environ.setdefault(k, v)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from spectrogram_midi_amd import _lib
from tools import signals

dev = torch.device("cuda", 0)
h = _lib.Handle()
h.set_profiling(True)
kinds = {"guitar": lambda s: signals.guitar_clip(120.0, seed=s), "polyphonic": lambda s: signals.polyphonic_clip(120.0, 44100, seed=s),
         "guitar under -12 dBFS noise": lambda s: signals.guitar_clip(120.0, seed=s, noise_dbfs=-12.0)}
for name, make in kinds.items():
    clips = [make(700 + i) for i in range(4)] * 4
    L = len(clips[0]); n = len(clips); F = n * (1 + L // 512)
    d_pcm = torch.from_numpy(np.concatenate(clips)).to(dev)
    off = (np.arange(n + 1) * L).astype(np.int64)
    outs = {"f0": torch.empty(F, dtype=torch.float64, device=dev), "voiced_flag": torch.empty(F, dtype=torch.uint8, device=dev),
            "voiced_prob": torch.empty(F, dtype=torch.float64, device=dev)}
    for _ in range(3):
        h.analyze_batch_device(d_pcm.data_ptr(), off, {k: v.data_ptr() for k, v in outs.items()}, sync=True, stages=_lib.STAGE_PYIN)
    st = h.viterbi_stats()
    print(json.dumps({"kind": name, "us_per_step": round(h.kernel_ms("viterbi") * 1e3 / (L // 512), 3), "voiced_fraction": round(float(outs["voiced_flag"].float().mean()), 3),
                      "list_only_rate": round(st["list_only"] / max(1, st["wave_steps"]), 4), "skipped": round(st["skipped"] / max(1, st["wave_steps"]), 4)}), flush=True)
