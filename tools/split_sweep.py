"""One 180 s clip (BASELINE configs[1]), device-resident: time-split segment length x warm-up sweep.
    python tools/split_sweep.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from spectrogram_midi_amd import _lib
from tools import signals

dev = torch.device("cuda", 0)
y = signals.guitar_clip(180.0, seed=3)
F = 1 + len(y) // 512
d_pcm = torch.from_numpy(y).to(dev)
off = np.array([0, len(y)], np.int64)
outs = {"f0": torch.empty(F, dtype=torch.float64, device=dev), "voiced_flag": torch.empty(F, dtype=torch.uint8, device=dev),
        "voiced_prob": torch.empty(F, dtype=torch.float64, device=dev), "rms": torch.empty(F, dtype=torch.float32, device=dev),
        "rake_mask": torch.empty(F, dtype=torch.uint8, device=dev)}
ptrs = {k: v.data_ptr() for k, v in outs.items()}
ref = None
for sl, wu in [(0, 128), (768, 128), (768, 256), (768, 384), (512, 128), (512, 256), (640, 256), (1024, 128), (1024, 256), (384, 256), (256, 256)]:
    os.environ["AEGIS_TIME_SPLIT"] = str(sl)
    os.environ["AEGIS_SPLIT_WARMUP"] = str(wu)
    h = _lib.Handle()
    h.set_profiling(True)
    ts = []
    for _ in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h.analyze_batch_device(d_pcm.data_ptr(), off, ptrs, sync=True)
        ts.append(time.perf_counter() - t0)
    got = {k: v.cpu().numpy() for k, v in outs.items()}
    if ref is None:
        ref = got
    same = all(np.array_equal(got[k], ref[k], equal_nan=True) for k in got)
    lk = h.debug_fetch("seg_lock") if sl else np.zeros(0)
    lkp = lk[lk > 0]
    print(json.dumps({"seglen": sl, "warmup": wu, "ms": round(float(np.median(ts[2:])) * 1e3, 3), "viterbi_ms": round(h.kernel_ms("viterbi"), 3),
                      "segments": h.param("last_split_segments"), "redo": h.param("split_flagged_clips"), "rounds": h.param("split_rounds"),
                      "lock_median": float(np.median(lkp)) if len(lkp) else None, "lock_max": int(lkp.max()) if len(lkp) else None,
                      "never_or_carried": int((lk < 0).sum()), "equal": bool(same)}))
    h.close()
