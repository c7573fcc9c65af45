"""Time-split Viterbi (viterbi.hip) against the sequential kernel on the latency-bound workloads:
BASELINE.json configs[1] (one 180 s clip), rank 0's shard of the 512-clip folder on 8 GPUs (64 ragged clips: what every
GPU of the driver's N = 8 run holds), the 64 x 180 s shard.  Prints one JSON line per workload.

    python tools/bench_split.py [single] [rank8] [shard]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from spectrogram_midi_amd import _lib, dist as adist
from tools import signals

which = sys.argv[1:] or ["single", "rank8", "shard"]
dev = torch.device("cuda", 0)


FORCE = os.environ.get("BENCH_SPLIT_FORCE")          # segment length: the time split forced on every pass (no planning rule, no back-off)


def run(clips, label, reps=5, modes=(("sequential", "0"), ("time-split", FORCE)), kinds=None):
    n = np.array([len(c) for c in clips], np.int64)
    off = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
    F = int((n // 512 + 1).sum())
    d_pcm = torch.from_numpy(np.concatenate(clips)).to(dev)
    outs = {"f0": torch.empty(F, dtype=torch.float64, device=dev), "voiced_flag": torch.empty(F, dtype=torch.uint8, device=dev),
            "voiced_prob": torch.empty(F, dtype=torch.float64, device=dev), "rms": torch.empty(F, dtype=torch.float32, device=dev),
            "rake_mask": torch.empty(F, dtype=torch.uint8, device=dev)}
    ptrs = {k: v.data_ptr() for k, v in outs.items()}
    res, ref = {}, None
    for mode in modes:
        name, env = mode[0], mode[1]
        extra = mode[2] if len(mode) > 2 else {}
        if env is None:
            os.environ.pop("AEGIS_TIME_SPLIT", None)
        else:
            os.environ["AEGIS_TIME_SPLIT"] = env
        for k in ("AEGIS_SPLIT_HYBRID", "AEGIS_HYBRID_PCT", "AEGIS_HYBRID_ROUNDS", "AEGIS_HYBRID_MIN_SEG"):
            os.environ.pop(k, None)
        os.environ.update(extra)
        h = _lib.Handle()
        h.set_profiling(True)
        ts = []
        for _ in range(reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            h.analyze_batch_device(d_pcm.data_ptr(), off, ptrs, sync=True)
            ts.append(time.perf_counter() - t0)
        ms = float(np.median(ts[1:])) * 1e3
        got = {k: v.cpu().numpy() for k, v in outs.items()}
        if ref is None:
            ref = got
        same = all(np.array_equal(got[k], ref[k], equal_nan=True) for k in got)
        res[name] = {"ms": round(ms, 3), "audio_s_per_s": round(float(n.sum()) / 44100 / (ms * 1e-3), 1),
                     "kernel_ms": {k: round(h.kernel_ms(k), 3) for k in ("frame", "pyin_obs", "viterbi", "finalize")},
                     "segments": h.param("last_split_segments"), "hybrid_step": h.param("last_hybrid_step"), "calls": reps + 1,
                     "all_ms": [round(x * 1e3, 2) for x in ts[1:]],
                     "clips_redone_sequentially": h.param("split_flagged_clips"), "of_which_never_locked": h.param("split_unlocked_clips"),
                     "outputs_equal_sequential": bool(same),
                     "verify": dict(zip(("frames", "tubes_opened", "tubes_recorded", "too_wide", "-", "closed_elsewhere", "open_at_exact_run", "too_deep", "max_depth", "oob_in_bound", "last_column_tie", "records_full", "tubes_resolved", "path_changed", "tubes_with_rail", "rail_frames"), (int(x) for x in h.debug_fetch("split_verify")[:16])))}
        if h.param("last_split_segments") > 0:
            lk = h.debug_fetch("seg_lock")
            lk = lk[lk != 0]
            res[name]["lock_on_steps"] = {"runs": int(len(lk)), "never": int((lk == -1).sum()), "carried_on": int((lk == -2).sum()), "median": float(np.median(lk[lk > 0])) if (lk > 0).any() else None,
                                          "p90": float(np.quantile(lk[lk > 0], 0.9)) if (lk > 0).any() else None, "max": int(lk.max()) if len(lk) else None,
                                          "over_256": int((lk > 256).sum())}
        if h.param("last_split_segments") > 0 and kinds is not None:
            fl = h.debug_fetch("split_flags")
            order = np.argsort(-n, kind="stable")                    # the pass takes its clips longest first
            res[name]["flagged"] = [(kinds[order[i]], round(float(n[order[i]]) / 44100), int(fl[i])) for i in range(len(fl)) if fl[i]]
        h.close()
    print(json.dumps({"workload": label, "clips": len(clips), "audio_s": round(float(n.sum()) / 44100, 1), "frames": F, **res}), flush=True)


if "single" in which:
    run([signals.guitar_clip(180.0, seed=1)], "configs[1]: one 180 s clip")
if "rank8" in which:
    durations = bench.folder_durations(512)
    mine = adist.shard_clips(durations, 8)[0]
    run(bench.make_folder_clips(mine, durations), "rank 0 of 8: its 64 clips of the 512-clip folder", kinds=[bench.FOLDER_KINDS[i % 8] for i in mine])
HYB = (("sequential", "0"), ("split, frame stage in front", FORCE, {"AEGIS_SPLIT_HYBRID": "0"}), ("hybrid", FORCE)) + tuple(
    (f"hybrid {pct} % / {r} round(s) / min {ms}", FORCE, {"AEGIS_HYBRID_PCT": str(pct), "AEGIS_HYBRID_ROUNDS": str(r), "AEGIS_HYBRID_MIN_SEG": str(ms)})
    for pct, r, ms in (tuple(int(x) for x in t.split(":")) for t in os.environ.get("BENCH_HYB", "100:3:768,100:4:512,100:5:512,100:6:384,108:4:512,92:4:512").split(",")))
if "rank8h" in which:           # the hybrid split pass (aegis_api.hip split_hybrid) on a rank's shard (BENCH_SPLIT_RANKS, default 0)
    durations = bench.folder_durations(512)
    for r in (int(x) for x in os.environ.get("BENCH_SPLIT_RANKS", "0").split(",")):
        mine = adist.shard_clips(durations, 8)[r]
        run(bench.make_folder_clips(mine, durations), f"rank {r} of 8, hybrid split pass", reps=6, modes=HYB, kinds=[bench.FOLDER_KINDS[i % 8] for i in mine])
if "ranksh" in which:
    durations = bench.folder_durations(512)
    shards = adist.shard_clips(durations, 8)
    for r in range(8):
        run(bench.make_folder_clips(shards[r], durations), f"rank {r} of 8", reps=8, modes=(HYB[0], HYB[1], HYB[2]), kinds=[bench.FOLDER_KINDS[i % 8] for i in shards[r]])
if "rank8auto" in which:        # rank R of 8 (BENCH_SPLIT_RANKS, default 0 and 4) under the automatic rule only
    durations = bench.folder_durations(512)
    shards = adist.shard_clips(durations, 8)
    for r in (int(x) for x in os.environ.get("BENCH_SPLIT_RANKS", "0,4").split(",")):
        run(bench.make_folder_clips(shards[r], durations), f"rank {r} of 8", reps=8, modes=(("sequential", "0"), ("automatic (hybrid)", None)),
            kinds=[bench.FOLDER_KINDS[i % 8] for i in shards[r]])
if "rank4h" in which:           # a rank of FOUR: 128 ragged clips, un-partitioned streams -- the planner's hybrid-only rule
    durations = bench.folder_durations(512)
    shards = adist.shard_clips(durations, 4)
    for r in (int(x) for x in os.environ.get("BENCH_SPLIT_RANKS", "0,3").split(",")):
        run(bench.make_folder_clips(shards[r], durations), f"rank {r} of 4", reps=6,
            modes=(("sequential", "0"), ("automatic (hybrid)", None), ("hybrid 85 %", None, {"AEGIS_HYBRID_PCT": "85"}), ("hybrid 115 %", None, {"AEGIS_HYBRID_PCT": "115"})),
            kinds=[bench.FOLDER_KINDS[i % 8] for i in shards[r]])
if "rank8tonal" in which:       # rank 0's shard with the noisy eighth of the folder replaced by tonal clips: what the split does when every clip locks on
    durations = bench.folder_durations(512)
    mine = adist.shard_clips(durations, 8)[0]
    kinds_save = bench.FOLDER_KINDS
    bench.FOLDER_KINDS = ("guitar",) * 6 + ("polyphonic", "guitar")
    run(bench.make_folder_clips(mine, durations), "rank 0 of 8, noisy clips replaced by tonal ones", reps=8, kinds=[bench.FOLDER_KINDS[i % 8] for i in mine])
    bench.FOLDER_KINDS = kinds_save
if "ranks" in which:            # every rank's shard of the folder at N = 8, one after the other on this GPU: what the driver's N = 8 run would see
    durations = bench.folder_durations(512)
    shards = adist.shard_clips(durations, 8)
    for r in range(8):
        run(bench.make_folder_clips(shards[r], durations), f"rank {r} of 8", reps=8, kinds=[bench.FOLDER_KINDS[i % 8] for i in shards[r]])
if "shard" in which:
    run(bench.make_clips(64, 180.0, seed0=1), "64 x 180 s", modes=(("sequential", "0"), ("forced 4096", "4096")), kinds=["guitar"] * 64)
