"""One rank shard of the folder (rank R of WORLD, default 0 of 8: python tools/split_probe.py [R [WORLD]]) through the device entry a few times; meant to run under
rocprofv3 --kernel-trace (tools/gpu_split_timeline.sh), environment decides the schedule (AEGIS_SPLIT_HYBRID, ...)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from spectrogram_midi_amd import _lib, dist as adist

r = int(sys.argv[1]) if len(sys.argv) > 1 else 0
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
durations = bench.folder_durations(512)
clips = bench.make_folder_clips(adist.shard_clips(durations, world)[r], durations)
dev = torch.device("cuda", 0)
n = np.array([len(c) for c in clips], np.int64)
off = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
F = int((n // 512 + 1).sum())
d_pcm = torch.from_numpy(np.concatenate(clips)).to(dev)
outs = {"f0": torch.empty(F, dtype=torch.float64, device=dev), "voiced_flag": torch.empty(F, dtype=torch.uint8, device=dev),
        "voiced_prob": torch.empty(F, dtype=torch.float64, device=dev), "rms": torch.empty(F, dtype=torch.float32, device=dev),
        "rake_mask": torch.empty(F, dtype=torch.uint8, device=dev)}
h = _lib.Handle()
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h.analyze_batch_device(d_pcm.data_ptr(), off, {k: v.data_ptr() for k, v in outs.items()}, sync=True)
    print(f"rank {r} of {world} ({len(clips)} clips, {F} frames): call {(time.perf_counter() - t0) * 1e3:.2f} ms, dense {h.param('last_dense')}, segments {h.param('last_split_segments')}, hybrid step {h.param('last_hybrid_step')}", flush=True)
h.close()
