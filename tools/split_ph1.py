"""Per-step time of the speculative runs (first launch of the split kernel) against the number of concurrent segments, on
tonal material (64 x 60 s guitar clips) and on rank 0's shard of the folder; run under rocprofv3 --kernel-trace and read
the launches' durations with tools/split_ph1.py --read <dir>.
    rocprofv3 --kernel-trace -d gpurun_out/ph1 --output-format csv -- python3 tools/split_ph1.py"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--read":
    rows = list(csv.DictReader(open(glob.glob(sys.argv[2] + "/*/*_kernel_trace.csv")[0])))
    k = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if "split_kernel" in r["Kernel_Name"]]
    plan = json.load(open(sys.argv[2] + "/plan.json"))
    i = 0
    for name, seglen, nseg, calls in plan:
        per = [k[i + 6 * c] for c in range(calls)]
        i += 6 * calls
        steps = seglen + 256
        print(f"{name:28s} seglen {seglen:6d} segments {nseg:4d}  speculative runs {min(per):7.2f} ms = {min(per) * 1e3 / steps:5.2f} us/step")
    sys.exit(0)
import numpy as np
import torch
import bench
from spectrogram_midi_amd import _lib, dist as adist
from tools import signals

dev = torch.device("cuda", 0)
plan = []


def run(clips, name, seglens):
    n = np.array([len(c) for c in clips], np.int64)
    off = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
    F = int((n // 512 + 1).sum())
    d_pcm = torch.from_numpy(np.concatenate(clips)).to(dev)
    outs = {"f0": torch.empty(F, dtype=torch.float64, device=dev), "voiced_flag": torch.empty(F, dtype=torch.uint8, device=dev),
            "voiced_prob": torch.empty(F, dtype=torch.float64, device=dev)}
    ptrs = {k: v.data_ptr() for k, v in outs.items()}
    for sl in seglens:
        os.environ["AEGIS_TIME_SPLIT"] = str(sl)
        h = _lib.Handle()
        for _ in range(3):
            h.analyze_batch_device(d_pcm.data_ptr(), off, ptrs, sync=True, stages=_lib.STAGE_PYIN)
        plan.append((name, sl // 16 * 16, h.param("last_split_segments"), 3))
        h.close()


base = [signals.guitar_clip(60.0, seed=50 + i) for i in range(8)]
run([base[i % 8] for i in range(64)], "64 x 60 s tonal", (5152, 2576, 1280, 640))
durations = bench.folder_durations(512)
mine = adist.shard_clips(durations, 8)[0]
run(bench.make_folder_clips(mine, durations), "rank 0 of the folder", (16320, 8160, 4080, 2032))
out = [a for a in sys.argv[1:] if a.startswith("--out=")]
json.dump(plan, open((out[0][6:] if out else "gpurun_out/ph1") + "/plan.json", "w"))
