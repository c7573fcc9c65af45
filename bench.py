#!/usr/bin/env python3
"""Throughput benchmark of the analyze hot path (BASELINE.json metric: audio-seconds
transcribed per second, 44.1 kHz mono, n_fft=2048, hop=512).

A step = one pass of the whole analyze path (mel/dB/rake + pYIN + RMS) over this rank's batch
of synthetic clips, PCM already resident in HBM, outputs left in HBM.  Workload: the per-GPU
shard of BASELINE.json configs[3] (512-clip folder over 8 GPUs = 64 clips per GPU) with clips
of the configs[1] shape (3-minute 44.1 kHz mono guitar clips).  Ranks are independent (weak
scaling, no data-path collective); rank 0 prints ONE JSON line.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SR, HOP = 44100, 512
# read the PCM once (4 B x 44100) + write the raw_data arrays
# (f0 8 + voiced 1 + prob 8 + rms 4 + rake 1 = 22 B x 86.13 frames) per audio-second
ALGO_BYTES_PER_AUDIO_S = 4 * SR + 22 * (SR / HOP)
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec


def make_clips(n_clips, seconds, seed0):
    """n_clips distinct synthetic guitar clips.  Eight base clips are synthesised (Karplus-Strong
    notes + rake bursts + noise floor, signals.guitar_clip); the rest are circular shifts of them
    with a different gain, which keeps set-up time bounded without repeating any clip."""
    from spectrogram_midi_amd import signals
    n_base = min(n_clips, 8)
    base = [signals.guitar_clip(seconds, SR, seed=seed0 + i) for i in range(n_base)]
    rng = np.random.default_rng(seed0)
    clips = []
    for i in range(n_clips):
        b = base[i % n_base]
        if i < n_base:
            clips.append(b)
        else:
            clips.append((np.roll(b, int(rng.integers(1, len(b)))) * np.float32(rng.uniform(0.5, 1.0))).astype(np.float32))
    return clips


def cpu_baseline(sample_seconds):
    """Times the CPU oracle (oracle/, a NumPy restatement of the reference's librosa path with a C
    Viterbi) on a bounded sample of the same workload, single process."""
    from oracle import engine as oracle_engine
    from spectrogram_midi_amd import signals
    y = signals.guitar_clip(sample_seconds, SR, seed=1)
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(k, "1")
    t0 = time.perf_counter()
    oracle_engine.audio_to_midi(y)
    dt = time.perf_counter() - t0
    return {"value": round(sample_seconds / dt, 3), "unit": "audio-seconds/s", "cores": 1, "kind": "port",
            "sample": f"one {sample_seconds:g} s clip of the bench workload (signals.guitar_clip seed 1), "
                      f"oracle.engine.audio_to_midi stable path, {dt:.1f} s wall",
            "host_cpus": os.cpu_count()}


def measured_traffic(kernel, args):
    """HBM bytes per step of `kernel` from the committed rocprofv3 --pmc passes (profiles/
    r1_pmc_hbm.json: FETCH_SIZE and WRITE_SIZE in KB, summed over the launches of one step;
    FETCH_SIZE doubled per MI355X_MICROARCH.md, HBM section).  Only valid for the default workload."""
    path = os.path.join(ROOT, "profiles", "r1_pmc_hbm.json")
    if args.clips != 64 or args.clip_seconds != 180.0 or not os.path.exists(path):
        return None, None
    with open(path) as f:
        pmc = json.load(f)
    rec = pmc.get("per_step", {}).get(kernel)
    if not rec:
        return None, None
    return int((2 * rec["FETCH_SIZE_KB"] + rec["WRITE_SIZE_KB"]) * 1024), "profiles/r1_pmc_hbm.json"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--clips", type=int, default=64, help="clips per GPU")
    ap.add_argument("--clip-seconds", type=float, default=180.0)
    ap.add_argument("--cpu-sample-seconds", type=float, default=240.0, help="oracle sample (about 15 s of CPU work)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="test hook: every rank uses cuda:0 and the gloo backend (multi-rank path on a 1-GPU box)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the analyze path has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = torch.device("cpu") if args.rehearse_on_one_gpu else dev     # where the timing reductions live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from spectrogram_midi_amd import _lib

    clips = make_clips(args.clips, args.clip_seconds, seed0=1000 * rank + 1)
    n_samples = np.array([len(c) for c in clips], dtype=np.int64)
    offsets = np.concatenate([[0], np.cumsum(n_samples)]).astype(np.int64)
    audio_seconds = float(n_samples.sum()) / SR
    handle = _lib.Handle(sample_rate=SR, hop_length=HOP, device=local_rank,
                         max_frames_per_pass=max(1 << 21, int(n_samples.sum() // HOP + len(n_samples) + 1)))
    frames = int(sum(handle.frames_for(int(n)) for n in n_samples))

    d_pcm = torch.from_numpy(np.concatenate(clips)).to(dev)
    d_out = {
        "f0": torch.empty(frames, dtype=torch.float64, device=dev),
        "voiced_flag": torch.empty(frames, dtype=torch.uint8, device=dev),
        "voiced_prob": torch.empty(frames, dtype=torch.float64, device=dev),
        "rms": torch.empty(frames, dtype=torch.float32, device=dev),
        "rake_mask": torch.empty(frames, dtype=torch.uint8, device=dev),
    }
    out_ptrs = {k: v.data_ptr() for k, v in d_out.items()}
    del clips

    def step():
        handle.analyze_batch_device(d_pcm.data_ptr(), offsets, out_ptrs, rake_sensitivity=0.6,
                                    stages=_lib.STAGE_ALL, sync=True)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    handle.set_profiling(True)
    kernel_ms, kernel_n = {}, {}
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        for k in ("frame_fft", "yin_seq", "pyin_obs", "viterbi", "finalize"):
            kernel_ms[k] = kernel_ms.get(k, 0.0) + handle.kernel_ms(k)
            kernel_n[k] = kernel_n.get(k, 0) + handle.kernel_launches(k)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([audio_seconds], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_audio = float(tot.item())
    else:
        total_audio = audio_seconds

    if rank == 0:
        kernel_ms = {k: v / args.steps for k, v in kernel_ms.items()}
        dom = max(kernel_ms, key=kernel_ms.get)
        dom_ms = kernel_ms[dom]
        achieved = ALGO_BYTES_PER_AUDIO_S * audio_seconds / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        launches = max(1, kernel_n.get(dom, 0) // max(1, args.steps))
        traffic, traffic_src = measured_traffic(dom, args)
        voiced = float(d_out["voiced_flag"].float().mean().item())
        line = {
            "metric": "audio-seconds transcribed/sec (44.1 kHz, n_fft=2048)",
            "value": round(total_audio * args.steps / elapsed, 2),
            "unit": "audio-seconds/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"configs[3] per-GPU shard: {args.clips} clips x {args.clip_seconds:g} s "
                                   "(configs[1] clip shape), full mel/dB/rake + pYIN + RMS",
                       "clips_per_gpu": args.clips, "clip_seconds": args.clip_seconds, "sample_rate": SR,
                       "n_fft": 2048, "hop_length": HOP, "frames_per_gpu": frames,
                       "parallelism": f"clips sharded over {world} GPU(s), no collective on the data path"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "traffic_source": traffic_src,
                         # the time-chunked pipeline launches the kernel once per time chunk: bytes and
                         # duration below are per step (= sum over those launches); avg_launch_ms is what
                         # rocprofv3 --stats reports as AverageNs
                         "launches_per_step": launches, "avg_launch_ms": round(dom_ms / launches, 3),
                         "algorithmic_bytes_per_step": int(ALGO_BYTES_PER_AUDIO_S * audio_seconds),
                         "algorithmic_bytes_per_launch": int(ALGO_BYTES_PER_AUDIO_S * audio_seconds / launches),
                         "traffic_per_launch": None if traffic is None else int(traffic / launches),
                         "kernel_ms": {k: round(v, 3) for k, v in kernel_ms.items()}},
            "voiced_fraction": round(voiced, 4),
        }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
