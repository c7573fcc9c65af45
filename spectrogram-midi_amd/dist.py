"""Multi-GPU layout of a folder of clips (BASELINE.json configs[3]): clips are
independent, so they are sharded over ranks with no collective on the data path -- one process per
GPU, longest-processing-time-first assignment -- and only the note events travel at the end
(`gather_events`, a torch.distributed gather: RCCL over xGMI on GPUs, gloo in the CPU tests).
The reference's only parallelism is a process pool over time chunks (aegis_engine.py:183-216)."""
import numpy as np

_TECHNIQUES = (None, "vibrato", "bend", "slide", "hammer_on", "pull_off")


def shard_clips(durations, world_size):
    """LPT assignment: returns a list (per rank) of clip indices.  Deterministic: ties go to the
    lower clip index and the lower rank."""
    order = sorted(range(len(durations)), key=lambda i: (-float(durations[i]), i))
    load = [0.0] * world_size
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda q: (load[q], q))
        shards[r].append(i)
        load[r] += float(durations[i])
    return shards


def pack_events(clip_index, events):
    """events (list of dicts, midi_logic schema) -> float64 [n, 10] rows:
    clip, note, start, end, velocity, track(0 main/1 safe), technique id, confidence, slope, rms_energy."""
    rows = np.zeros((len(events), 10), np.float64)
    for i, e in enumerate(events):
        rows[i] = (clip_index, e["note"], e["start"], e["end"], e["velocity"], 0 if e["track"] == "main" else 1,
                   _TECHNIQUES.index(e.get("technique")), float(e["confidence"]), float(e.get("slope", 0.0)),
                   float(e["rms_energy"]))
    return rows


def rows_from_packed(events, clip_ids):
    """The structured event array of events_native.extract_batch(packed=True) (field `clip` = index into the batch)
    -> the float64 [n, 10] rows of pack_events, with the batch indices replaced by the global clip ids."""
    ev = np.asarray(events)
    rows = np.zeros((len(ev), 10), np.float64)
    if len(ev):
        rows[:, 0] = np.asarray(clip_ids)[ev["clip"]]
        for j, k in enumerate(("note", "start", "end", "velocity"), start=1):
            rows[:, j] = ev[k]
        rows[:, 5] = 1 - ev["track"].astype(np.int64)            # native: 1 main / 0 safe; rows: 0 main / 1 safe
        rows[:, 6] = ev["technique"]
        rows[:, 7], rows[:, 8], rows[:, 9] = ev["confidence"], ev["slope"], ev["rms_energy"]
    return rows


def unpack_events(rows):
    """Inverse of pack_events -> {clip_index: [event dict, ...]} (events keep their order)."""
    out = {}
    for r in np.asarray(rows).reshape(-1, 10):
        out.setdefault(int(r[0]), []).append({
            "note": int(r[1]), "start": int(r[2]), "end": int(r[3]), "confidence": float(r[7]),
            "velocity": int(r[4]), "track": "main" if int(r[5]) == 0 else "safe", "rms_energy": float(r[9]),
            "technique": _TECHNIQUES[int(r[6])], "slope": float(r[8])})
    return out


def gather_events(local_rows, dst=0, device=None):
    """Gathers every rank's packed event rows on `dst` (variable lengths: sizes first, then one
    padded all_gather).  Returns the concatenated rows on dst, None elsewhere."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    local = torch.as_tensor(np.asarray(local_rows, np.float64).reshape(-1, 10), device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([local.shape[0]], dtype=torch.int64, device=dev))
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    padded = torch.zeros((cap, 10), dtype=torch.float64, device=dev)
    padded[: local.shape[0]] = local
    bufs = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded)
    if rank != dst:
        return None
    return np.concatenate([b[:n].cpu().numpy() for b, n in zip(bufs, counts)], axis=0)
