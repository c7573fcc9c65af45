"""Verification / exact-walk kernel time by material (time split forced, 8 clips x 180 s of one kind per call); run under
rocprofv3 --kernel-trace and read with --read <dir>."""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--read":
    rows = list(csv.DictReader(open(glob.glob(sys.argv[2] + "/*/*_kernel_trace.csv")[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ex = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if "exact_kernel" in r["Kernel_Name"]]
    ve = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if "verify_kernel" in r["Kernel_Name"]]
    plan = json.load(open(sys.argv[2] + "/plan.json"))
    i = 0
    for name, calls, tubes in plan:
        print(f"{name:30s} verify {min(ve[i:i + calls]):6.2f} ms  exact walk {min(ex[i:i + calls]):6.2f} ms  tubes per call {tubes}")
        i += calls
    sys.exit(0)
os.environ["AEGIS_TIME_SPLIT"] = "1536"
import numpy as np
from spectrogram_midi_amd import _lib
from tools import signals

kinds = {"guitar": lambda s: signals.guitar_clip(180.0, seed=s), "polyphonic": lambda s: signals.polyphonic_clip(180.0, 44100, seed=s),
         "guitar under -12 dBFS noise": lambda s: signals.guitar_clip(180.0, seed=s, noise_dbfs=-12.0)}
plan = []
for name, make in kinds.items():
    clips = [make(700 + i) for i in range(8)]
    h = _lib.Handle()
    for _ in range(3):
        h.analyze_batch(clips, stages=_lib.STAGE_PYIN)
    v = h.debug_fetch("split_verify")
    plan.append((name, 3, int(v[1]) // 3))
    h.close()
json.dump(plan, open(sys.argv[1] + "/plan.json", "w"))
