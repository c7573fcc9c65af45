"""bench.py's multi-rank contract: `python bench.py --gpus N` starts the N rank processes itself (fresh children, the
parent never touches the GPU), a launcher/--gpus mismatch is an error, and the folder workload is the one BASELINE.json
configs[3] describes.  The GPU test runs the bare --gpus 2 form on the one-GPU box (both ranks on cuda:0, gloo)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench   # noqa: E402


def test_gpus_flag_must_match_the_launcher():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_parent_launcher_does_not_import_torch_and_propagates_failure():
    """Without a GPU every rank exits non-zero; the parent must start them, not touch the GPU itself, and report it."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def launch_ranks")]
    assert "import torch" not in head                      # torch is only imported inside the rank
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""                        # ranks see no device here or on the GPU box
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "needs a GPU" in r.stderr


def test_folder_workload_matches_configs3():
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        a = bench.parse_args()
    finally:
        sys.argv = argv
    assert a.config == "folder" and a.folder_clips == 512 and a.gpus == 1      # the driver's bare run = configs[3] as written
    d = bench.folder_durations(512)
    assert len(d) == 512 and d.min() >= 30 and d.max() <= 330 and abs(d.mean() - 180) < 10
    np.testing.assert_array_equal(d, bench.folder_durations(512))        # seeded
    from spectrogram_midi_amd import dist
    shards = dist.shard_clips(d, 8)
    assert all(len(s) >= 50 for s in shards)
    loads = [d[s].sum() for s in shards]
    assert (max(loads) - min(loads)) / np.mean(loads) < 0.02              # longest-first keeps the ranks level
    clips = bench.make_folder_clips([6, 7, 8], np.array([1.0] * 9))      # polyphonic, noisy, guitar
    assert [len(c) for c in clips] == [44100] * 3 and all(c.dtype == np.float32 for c in clips)
    assert np.std(clips[1]) > np.std(clips[2]) * 0.5


@pytest.mark.gpu
def test_bare_gpus_2_runs_two_ranks_and_gathers_events():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--config", "shard",
           "--clips", "3", "--clip-seconds", "8", "--no-cpu-baseline", "--rehearse-on-one-gpu"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                               # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and len(line["rank_busy_ms"]) == 2
    assert line["events"]["backend"] == "gloo" and line["events"]["gather_ms"] is not None and line["events"]["count"] > 0
    assert abs(sum(line["rank_audio_seconds"]) - 2 * 3 * 8) < 0.1
    # sub-records of the same run: the host-buffer entry and the engine surface, whole-job rates
    assert line["host_inclusive"]["value"] > 0 and line["engine_e2e"]["value"] > 0 and line["engine_e2e"]["midi_bytes"] > 0
    assert line["engine_e2e"]["per_clip_api"]["audio_to_midi_ms"] >= line["engine_e2e"]["per_clip_api"]["analyze_ms"] and line["uniform_shard"] is None


@pytest.mark.gpu
def test_folder_mode_two_ranks():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--config", "folder",
           "--folder-clips", "6", "--no-cpu-baseline", "--rehearse-on-one-gpu"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    d = bench.folder_durations(6)
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert abs(line["config"]["folder_audio_seconds"] - float(np.floor(d * 44100).sum() / 44100)) < 0.1
    assert 0.0 <= line.get("viterbi_list_only_rate", 0.5) <= 1.0
    assert "folder of 6 clips" in line["config"]["workload"] and line["host_inclusive"]["value"] > 0
    # the two ranks' outputs, digested per clip, are the outputs of the six clips on one handle in one process
    from spectrogram_midi_amd import _lib
    clips = bench.make_folder_clips(range(6), d)
    h = _lib.Handle()
    res = h.analyze_batch(clips, want_sdb=False)
    h.close()
    host = {k: np.concatenate([r[k] for r in res]) for k in ("f0", "voiced_flag", "voiced_prob", "rms", "rake_mask")}
    f_off = np.concatenate([[0], np.cumsum([len(r["f0"]) for r in res])])
    assert line["outputs_check"]["digest"] == bench.outputs_digest(host, f_off, range(6))
    assert line["outputs_check"]["expected"] is None and line["outputs_check"]["match"] is None      # not the 512-clip folder


def test_outputs_digest_does_not_depend_on_the_sharding():
    """bench.outputs_digest (the bench line's `outputs_check`): per-clip CRCs seeded with the clip's folder index, summed --
    the same number however dist.shard_clips deals the clips to 1, 2, 4 or 8 ranks, different when one frame of one clip
    changes or two clips swap their outputs."""
    from spectrogram_midi_amd import dist
    rng = np.random.default_rng(3)
    n = 40
    frames = rng.integers(1, 400, n)
    per_clip = [{"f0": np.where(rng.random(f) < 0.3, np.nan, rng.uniform(80, 900, f)), "voiced_flag": (rng.random(f) < 0.7).astype(np.uint8),
                 "rake_mask": (rng.random(f) < 0.1).astype(np.uint8), "rms": rng.random(f).astype(np.float32), "voiced_prob": rng.random(f)}
                for f in frames]

    def digest(world, clips=per_clip):
        total = 0
        for mine in dist.shard_clips(frames.astype(float), world):
            host = {k: np.concatenate([clips[i][k] for i in mine]) if len(mine) else np.zeros(0) for k in per_clip[0]}
            off = np.concatenate([[0], np.cumsum([frames[i] for i in mine])]).astype(np.int64)
            total += bench.outputs_digest(host, off, list(mine))
        return total

    d1 = digest(1)
    assert d1 == digest(2) == digest(4) == digest(8) and d1 < 2 ** 53          # (summed over ranks in float64 by bench.py)
    changed = [dict(c) for c in per_clip]
    changed[7]["rms"] = changed[7]["rms"].copy()
    changed[7]["rms"][0] += np.float32(1e-3)
    assert digest(4, changed) != d1
    same_len = [i for i in range(n) if frames[i] == frames[0]]
    swapped = list(per_clip)
    j = same_len[1] if len(same_len) > 1 else None
    if j is not None:
        swapped[0], swapped[j] = swapped[j], swapped[0]
        assert digest(2, swapped) != d1
