"""The band Viterbi alone on the chip (one time chunk: frame stage first, then every clip's recurrence), for growing numbers
of clips: microseconds per step as the workgroups fill the compute units.
    AEGIS_TIME_CHUNK=100000000 AEGIS_CU_SPLIT=0 AEGIS_TIME_SPLIT=0 AEGIS_DENSE=0 python tools/viterbi_alone.py"""
import json, os, sys
for k, v in (("AEGIS_TIME_CHUNK", "100000000"), ("AEGIS_CU_SPLIT", "0"), ("AEGIS_TIME_SPLIT", "0"), ("AEGIS_DENSE", "0"), ("AEGIS_PROPORTIONAL_CHUNKS", "0")):
    os.environ.setdefault(k, v)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from spectrogram_midi_amd import _lib
from tools import signals

dev = torch.device("cuda", 0)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
base = [signals.guitar_clip(secs, seed=50 + i) for i in range(8)]
h = _lib.Handle()
h.set_profiling(True)
for n in (16, 32, 64, 96, 128, 160, 192, 224, 256, 288, 384, 512):
    clips = [base[i % 8] for i in range(n)]
    L = len(clips[0])
    F = n * (1 + L // 512)
    d_pcm = torch.from_numpy(np.concatenate(clips)).to(dev)
    off = (np.arange(n + 1) * L).astype(np.int64)
    outs = {"f0": torch.empty(F, dtype=torch.float64, device=dev), "voiced_flag": torch.empty(F, dtype=torch.uint8, device=dev),
            "voiced_prob": torch.empty(F, dtype=torch.float64, device=dev)}
    ptrs = {k: v.data_ptr() for k, v in outs.items()}
    for _ in range(3):
        h.analyze_batch_device(d_pcm.data_ptr(), off, ptrs, sync=True, stages=_lib.STAGE_PYIN)
    steps = L // 512
    print(json.dumps({"clips": n, "steps": steps, "viterbi_ms": round(h.kernel_ms("viterbi"), 3), "launches": h.kernel_launches("viterbi"), "chunks": h.param("last_chunks"),
                      "us_per_step": round(h.kernel_ms("viterbi") * 1e3 / steps, 3), "frame_ms": round(h.kernel_ms("frame"), 3), "obs_ms": round(h.kernel_ms("pyin_obs"), 3)}), flush=True)
