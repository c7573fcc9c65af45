"""world_size-2 gloo test of the N>1 path: clips sharded per rank with no data-path collective,
note events gathered on rank 0 (dist.gather_events).  The per-rank analysis is stood in for by
precomputed oracle raw_data (no GPU here); what is under test is sharding + gather."""
import os
import socket

import numpy as np
import torch.distributed as tdist
import torch.multiprocessing as mp

from spectrogram_midi_amd import dist
from tools import signals


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, durations, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    mine = dist.shard_clips(durations, world)[rank]
    rows = []
    for ci in mine:     # deterministic stand-in events: one note per clip second
        ev = [{"note": 40 + (ci + k) % 40, "start": 10 * k, "end": 10 * k + 5, "confidence": 0.5 + 0.01 * k,
               "velocity": 64 + k, "track": "main" if k % 2 else "safe", "rms_energy": -3.0 * k,
               "technique": (None, "vibrato", "hammer_on")[k % 3], "slope": 0.1 * k} for k in range(int(durations[ci]))]
        rows.append(dist.pack_events(ci, ev))
    local = np.concatenate(rows) if rows else np.zeros((0, 10))
    out = dist.gather_events(local, dst=0)
    if rank == 0:
        ret["rows"] = out
    else:
        assert out is None
    tdist.barrier()
    tdist.destroy_process_group()


def test_two_rank_shard_and_gather():
    durations = [3.0, 9.0, 1.0, 4.0, 7.0, 2.0, 0.0]
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(2, _free_port(), durations, ret), nprocs=2, join=True)
        rows = ret["rows"]
    per_clip = dist.unpack_events(rows)
    assert sorted(per_clip) == [0, 1, 2, 3, 4, 5]                 # clip 6 has no events
    for ci, ev in per_clip.items():
        assert len(ev) == int(durations[ci])
        assert [e["start"] for e in ev] == [10 * k for k in range(len(ev))]
        assert ev[0]["technique"] is None and (len(ev) < 2 or ev[1]["technique"] == "vibrato")


def test_eight_ranks_with_empty_ranks_and_eventless_ranks():
    """world_size 8 (the node the driver's scaling run uses): five clips over eight ranks leave three ranks without a clip,
    and one rank's only clip has no events -- the padded all_gather must carry zero-row contributions, and the LPT
    shards must be disjoint, complete and longest-first."""
    durations = [5.0, 0.0, 3.0, 8.0, 2.0]
    shards = dist.shard_clips(durations, 8)
    assert sorted(i for s in shards for i in s) == list(range(5)) and sum(1 for s in shards if not s) == 3
    assert shards[0] == [3] and shards[1] == [0]
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(8, _free_port(), durations, ret), nprocs=8, join=True)
        rows = ret["rows"]
    per_clip = dist.unpack_events(rows)
    assert sorted(per_clip) == [0, 2, 3, 4] and rows.shape == (18, 10)
    for ci, ev in per_clip.items():
        assert len(ev) == int(durations[ci]) and [e["velocity"] for e in ev] == [64 + k for k in range(len(ev))]
    big = dist.shard_clips(list(np.random.default_rng(0).uniform(30, 330, 512)), 8)
    loads = [sum(np.random.default_rng(0).uniform(30, 330, 512)[i] for i in s) for s in big]
    assert max(loads) - min(loads) < 330.0 and all(len(s) >= 60 for s in big)
