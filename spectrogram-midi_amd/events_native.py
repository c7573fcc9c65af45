"""Note events and Standard MIDI Files for a batch of clips through the library's host code
(include/aegis_hip.h: aegis_extract_events, aegis_render_smf; csrc/events.cpp) -- the batched form of
`midi_logic.get_midi_events` (/root/reference/aegis_engine_core/midi_logic.py:32-148) and of the SMF block of
`AegisEngine.extract_events` (/root/reference/aegis_engine.py:98-179).

The two logarithms of the reference stay with NumPy (its float32 log10 / float64 log2 kernels are what the reference
runs): `amplitude_to_db(rms, ref=np.max)` and `hz_to_midi(f0)` are evaluated here for the whole batch in a handful of
array calls, everything else -- gating is a mask, then runs, minimum duration, articulation fits, merging, hammer-on /
pull-off tagging, MIDI bytes -- runs in C++ over the clips in parallel.  A clip with an articulation decision within
1e-9 of a threshold is decided by `midi_logic.detect_articulations` (the reference's own np.polyfit arithmetic) for
that note only, and the batch is run again with those verdicts.  Output: the reference's list-of-dicts schema, identical to the per-clip path."""
import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _lib, midi_logic, smf

try:        # csrc/pyevents.c, built by csrc/Makefile next to libaegis_hip.so
    from . import _aegis_pyevents as _pyevents
except ImportError:
    _pyevents = None

TECH = (None, "vibrato", "bend", "slide", "hammer_on", "pull_off")
_TECH_CODE = {t: i for i, t in enumerate(TECH)}


class Event(C.Structure):
    _fields_ = [("clip", C.c_int32), ("note", C.c_int32), ("start", C.c_int32), ("end", C.c_int32), ("velocity", C.c_int32),
                ("track", C.c_uint8), ("technique", C.c_uint8), ("reserved0", C.c_uint8), ("reserved1", C.c_uint8),
                ("rms_energy", C.c_float), ("reserved2", C.c_int32), ("confidence", C.c_double), ("slope", C.c_double)]


EVENT_DTYPE = np.dtype([("clip", "<i4"), ("note", "<i4"), ("start", "<i4"), ("end", "<i4"), ("velocity", "<i4"),
                        ("track", "u1"), ("technique", "u1"), ("reserved0", "u1"), ("reserved1", "u1"),
                        ("rms_energy", "<f4"), ("reserved2", "<i4"), ("confidence", "<f8"), ("slope", "<f8")])
assert EVENT_DTYPE.itemsize == C.sizeof(Event) == 48


class RunFit(C.Structure):
    _fields_ = [("clip", C.c_int32), ("start", C.c_int32), ("end", C.c_int32), ("technique", C.c_int32), ("slope", C.c_double)]


FIT_DTYPE = np.dtype([("clip", "<i4"), ("start", "<i4"), ("end", "<i4"), ("technique", "<i4"), ("slope", "<f8")])
assert FIT_DTYPE.itemsize == C.sizeof(RunFit) == 24


class EventBatch(C.Structure):
    _fields_ = [("n_clips", C.c_int32), ("reserved", C.c_int32), ("frame_off", C.c_void_p), ("sounding", C.c_void_p),
                ("semitones", C.c_void_p), ("pitch_bin", C.c_void_p), ("bin_semitones", C.c_void_p), ("rms_db", C.c_void_p),
                ("probs", C.c_void_p), ("fits", C.c_void_p), ("n_fits", C.c_int64)]


class EventParams(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("hop_length", C.c_int32), ("confidence_threshold", C.c_double),
                ("sustain_ms", C.c_double), ("min_note_duration_ms", C.c_double)]


def _bind():
    lib = _lib.load()
    if not hasattr(lib, "_events_bound"):
        lib.aegis_extract_events.argtypes = [C.POINTER(EventParams), C.POINTER(EventBatch), C.c_void_p, C.c_int64, C.c_void_p,
                                             C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        lib.aegis_extract_events.restype = C.c_int64
        lib.aegis_render_smf.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        lib.aegis_render_smf.restype = C.c_int64
        lib.aegis_events_last_error.restype = C.c_char_p
        lib._events_bound = True
    return lib


def _rms_db_block(rms, frame_off, amin=1e-5, top_db=80.0):
    """amplitude_to_db(rms, ref=np.max) of every clip of one clip-aligned block (frame_off relative to the block)."""
    counts = np.diff(frame_off)
    live = counts > 0
    starts = np.asarray(frame_off[:-1])[live]
    cl = counts[live]
    db = np.abs(rms)
    peak = np.maximum.reduceat(db, starts)                       # per live clip
    np.square(db, out=db)
    np.maximum(amin ** 2, db, out=db)
    np.log10(db, out=db)
    np.multiply(10.0, db, out=db)
    ref = 10.0 * np.log10(np.maximum(amin ** 2, np.square(peak)))
    db -= np.repeat(ref, cl)
    floor = np.maximum.reduceat(db, starts) - top_db
    return np.maximum(db, np.repeat(floor, cl), out=db)


_BLOCK_FRAMES = 1 << 17        # half a megabyte of float32 per temporary: stays in the allocator's and the core's cache


def _clip_blocks(frame_off, block=_BLOCK_FRAMES):
    """Cut clips 0..n into runs of whole clips of about `block` frames: [(c0, c1), ...]."""
    n = len(frame_off) - 1
    cuts = [0]
    while cuts[-1] < n:
        c0 = cuts[-1]
        c1 = int(np.searchsorted(frame_off, frame_off[c0] + block, side="right")) - 1
        cuts.append(min(n, max(c1, c0 + 1)))
    return list(zip(cuts[:-1], cuts[1:]))


def _host_workers():
    try:
        return max(1, min(16, len(os.sched_getaffinity(0))))
    except AttributeError:
        return max(1, min(16, os.cpu_count() or 1))


def _map_blocks(fn, blocks):
    """fn over the blocks on fresh threads, joined before return (NumPy's array kernels release the GIL)."""
    nw = min(_host_workers(), len(blocks))
    if nw <= 1:
        for b in blocks:
            fn(b)
        return
    with ThreadPoolExecutor(max_workers=nw) as pool:
        for _ in pool.map(fn, blocks):
            pass


def batch_rms_db(rms, frame_off, amin=1e-5, top_db=80.0, gate=None):
    """convert.amplitude_to_db_max (librosa.amplitude_to_db(rms, ref=np.max), midi_logic.py:51) for every clip of a
    concatenated float32 track: the same elementwise NumPy kernels on the same values as the per-clip form, the
    per-clip maxima through ufunc.reduceat -- evaluated over blocks of whole clips (temporaries of half a megabyte
    instead of one array of the whole folder per operation) on the host threads granted to the process.
    gate = (out bool array, fn(db block, a, b) -> bool block): a per-frame mask formed from each block while it is hot."""
    rms = np.asarray(rms)
    frame_off = np.asarray(frame_off, dtype=np.int64)
    if rms.size == 0:
        return np.zeros(0, rms.dtype)
    out = np.empty(rms.shape, rms.dtype if rms.dtype.kind == "f" else np.float64)

    def one(block):
        c0, c1 = block
        a, b = int(frame_off[c0]), int(frame_off[c1])
        if b > a:
            out[a:b] = _rms_db_block(rms[a:b], frame_off[c0:c1 + 1] - a, amin, top_db)
            if gate is not None:
                gate[0][a:b] = gate[1](out[a:b], a, b)

    _map_blocks(one, _clip_blocks(frame_off))
    return out


def extract_batch(frame_off, rake_mask, f0, voiced_flag, active_probs, rms, sr, hop_length, confidence_threshold=0.70,
                  want_midi=False, midi_program=27, vibrato_rate=5.0, vibrato_depth=0.3, packed=False, pitch_bin=None,
                  freqs=None, **kwargs):
    """Events (and SMF bytes) of every clip of a concatenated batch: clip c = frames frame_off[c] .. frame_off[c+1].
    -> list of event lists (the reference's dicts), or (events, [bytes per clip]) with want_midi.
    packed=True returns the structured array of all events + offsets instead of dicts (dist.pack_events's input).
    pitch_bin + freqs (the analysis's decoded bins, f0 == freqs[bin] where voiced): hz_to_midi comes from a table of
    len(freqs) entries instead of a logarithm per frame -- the same values."""
    lib = _bind()
    noise_gate_db = kwargs.get("noise_gate_db", -40)
    frame_off = np.ascontiguousarray(frame_off, dtype=np.int64)
    n = len(frame_off) - 1
    f0 = np.asarray(f0)
    rms = np.asarray(rms)
    probs = np.ascontiguousarray(active_probs, dtype=np.float64)
    on_grid = pitch_bin is not None and freqs is not None
    voiced = np.asarray(voiced_flag, bool)
    rake = np.asarray(rake_mask, bool)
    sounding = np.zeros(len(f0), bool)
    if on_grid:        # voiced <=> bin >= 0 <=> f0 = freqs[bin] > 0
        pitch_bin = np.ascontiguousarray(pitch_bin, dtype=np.int16)
        gate = lambda db, a, b: voiced[a:b] & ~(db < noise_gate_db) & ~rake[a:b]
    else:
        gate = lambda db, a, b: voiced[a:b] & ~(db < noise_gate_db) & (f0[a:b] > 0) & ~rake[a:b]
    rms_db = batch_rms_db(rms, frame_off, gate=(sounding, gate))
    if on_grid:
        bin_semi = np.ascontiguousarray(midi_logic.hz_to_midi(np.asarray(freqs, np.float64)))
        semitones = None
    else:
        semitones = np.zeros(len(f0))
        if sounding.any():
            semitones[sounding] = midi_logic.hz_to_midi(f0[sounding])
    snd = np.ascontiguousarray(sounding).view(np.uint8)
    par = EventParams(int(sr), int(hop_length), float(confidence_threshold), float(kwargs.get("sustain_ms", 50)),
                      float(kwargs.get("min_note_duration_ms", 50)))
    ev_off = np.zeros(n + 1, np.int64)
    batch = EventBatch(n, 0, frame_off.ctypes.data, snd.ctypes.data, None if on_grid else semitones.ctypes.data,
                       pitch_bin.ctypes.data if on_grid else None, bin_semi.ctypes.data if on_grid else None,
                       rms_db.ctypes.data, probs.ctypes.data, None, 0)
    cap = max(64, len(f0) // 32)
    rcap = 256
    fits = None
    while True:
        events = np.empty(cap, EVENT_DTYPE)
        risky = np.empty(rcap, FIT_DTYPE)
        n_risky = C.c_int64(0)
        total = lib.aegis_extract_events(C.byref(par), C.byref(batch), events.ctypes.data, cap, ev_off.ctypes.data,
                                         risky.ctypes.data, rcap, C.byref(n_risky))
        if total < 0:
            raise _lib.AegisError(int(total), lib.aegis_events_last_error().decode())
        if n_risky.value > rcap:
            rcap = int(n_risky.value)
            continue
        if n_risky.value > 0:
            # notes whose articulation decision sits within 1e-9 of a threshold: the reference's own arithmetic decides
            # (np.polyfit, midi_logic.detect_articulations) for those runs only, and the batch is run again with the verdicts
            if fits is not None:
                raise RuntimeError("native event extraction did not accept the supplied verdicts")
            fits = risky[:n_risky.value].copy()
            seen = {}       # the verdict is a function of the run's f0 values alone, and tracks on a pitch grid repeat their runs
            for r in fits:
                a = int(frame_off[r["clip"]])
                run = f0[a + int(r["start"]):a + int(r["end"]) + 1]
                key = run.tobytes()
                hit = seen.get(key)
                if hit is None:
                    tech, slope = midi_logic.detect_articulations(run, 0, len(run) - 1, sr, hop_length)
                    hit = seen[key] = (_TECH_CODE[tech], slope)
                r["technique"], r["slope"] = hit
            batch.fits, batch.n_fits = fits.ctypes.data, len(fits)
            continue
        if total <= cap:
            break
        cap = int(total)
    events = events[:total]
    blobs = None
    if want_midi:
        byte_off = np.zeros(n + 1, np.int64)
        bcap = 64 * (n + 1) + 40 * int(total) + 256 * int(np.count_nonzero(events["technique"] == 1) + np.count_nonzero(events["technique"] == 2))
        while True:
            out = np.empty(bcap, np.uint8)
            size = lib.aegis_render_smf(int(sr), int(hop_length), int(midi_program), float(vibrato_rate), float(vibrato_depth),
                                        n, events.ctypes.data, ev_off.ctypes.data, out.ctypes.data, bcap, byte_off.ctypes.data)
            if size < 0:
                raise ValueError(lib.aegis_events_last_error().decode())
            if size <= bcap:
                break
            bcap = int(size)
        raw = out[:size].tobytes()
        blobs = [raw[byte_off[c]:byte_off[c + 1]] for c in range(n)]
    if packed:
        return (events, ev_off, blobs) if want_midi else (events, ev_off)
    # ---- the reference's list of dicts ------------------------------------------------------------------------------
    conf = events["confidence"]                         # np.float64 / np.float32 scalars, as the reference's events carry
    energy = events["rms_energy"]
    if _pyevents is not None:       # the same dicts (same keys in the same order, same value types), built in C
        rows = _pyevents.event_dicts(np.ascontiguousarray(events), list(conf), list(energy), TECH)
        per_clip = [rows[ev_off[c]:ev_off[c + 1]] for c in range(n)]
        return (per_clip, blobs) if want_midi else per_clip
    rows = [{"note": a, "start": b, "end": c, "confidence": d, "velocity": e, "track": "main" if f else "safe",
             "rms_energy": g, "technique": TECH[h], "slope": i}
            for a, b, c, d, e, f, g, h, i in zip(events["note"].tolist(), events["start"].tolist(), events["end"].tolist(),
                                                 conf, events["velocity"].tolist(), events["track"].tolist(), energy,
                                                 events["technique"].tolist(), events["slope"].tolist())]
    per_clip = [rows[ev_off[c]:ev_off[c + 1]] for c in range(n)]
    return (per_clip, blobs) if want_midi else per_clip


def pack(per_clip):
    """list of event lists -> structured array (clip index filled in)."""
    total = sum(len(e) for e in per_clip)
    out = np.zeros(total, EVENT_DTYPE)
    k = 0
    for c, evs in enumerate(per_clip):
        for e in evs:
            out[k] = (c, e["note"], e["start"], e["end"], e["velocity"], 1 if e["track"] == "main" else 0,
                      _TECH_CODE[e.get("technique")], 0, 0, e["rms_energy"], 0, e["confidence"], e.get("slope", 0.0))
            k += 1
    return out


def get_midi_events(rake_mask, f0, voiced_flag, active_probs, rms, sr, hop_length, confidence_threshold, **kwargs):
    """midi_logic.get_midi_events through the native path (one clip)."""
    return extract_batch([0, len(f0)], rake_mask, f0, voiced_flag, active_probs, rms, sr, hop_length,
                         confidence_threshold, **kwargs)[0]


def render_smf(events, sr, hop_length, midi_program=27, vibrato_rate=5.0, vibrato_depth=0.3):
    """smf.render through the native writer (one clip's list of event dicts -> SMF bytes)."""
    lib = _bind()
    ev = pack([events])
    ev_off = np.array([0, len(ev)], np.int64)
    byte_off = np.zeros(2, np.int64)
    cap = 128 + 400 * max(1, len(ev))
    out = np.empty(cap, np.uint8)
    size = lib.aegis_render_smf(int(sr), int(hop_length), int(midi_program), float(vibrato_rate), float(vibrato_depth), 1,
                                ev.ctypes.data, ev_off.ctypes.data, out.ctypes.data, cap, byte_off.ctypes.data)
    if size < 0:
        raise ValueError(lib.aegis_events_last_error().decode())
    assert size <= cap
    return out[:size].tobytes()
