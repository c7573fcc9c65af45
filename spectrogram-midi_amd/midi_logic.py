"""Frame arrays -> note events: the host-side consumer of the analyze kernels.

Mirrors `get_midi_events` / `detect_articulations` of the reference
(/root/reference/aegis_engine_core/midi_logic.py:6-30, 32-148): same arguments, same event
dict schema, same quirks.  Written as array passes instead of the
reference's per-frame loop: gating and pitch quantisation are vectorised, notes are the runs of
equal pitch between change points, and only the per-note steps (articulation fit, merge,
hammer-on/pull-off tagging) iterate -- over notes, not frames.
"""
import numpy as np

from .convert import amplitude_to_db_max, hz_to_midi


def detect_articulations(f0, start, end, sr, hop_length):
    """(technique, slope) for frames start..end inclusive -- midi_logic.py:6-30."""
    if end <= start:
        return (None, 0.0)
    voiced = f0[start:end + 1]
    voiced = voiced[voiced > 0]
    n = len(voiced)
    if n < 3:
        return (None, 0.0)
    semitones = hz_to_midi(voiced)
    x = np.arange(n)
    coef = np.polyfit(x, semitones, 1)
    slope = coef[0]
    wobble = semitones - np.polyval(coef, x)
    if np.max(wobble) - np.min(wobble) > 0.3:
        return ("vibrato", slope)
    if slope > 0.05:
        return ("bend", slope)
    if abs(slope) > 0.02:
        return ("slide", slope)
    return (None, 0.0)


def _note_runs(sounding, pitch):
    """Inclusive (start, end) of every maximal run of sounding frames with one pitch."""
    n = len(sounding)
    if n == 0:
        return np.zeros(0, int), np.zeros(0, int)
    key = np.where(sounding, pitch, np.iinfo(np.int64).min)
    change = np.flatnonzero(key[1:] != key[:-1]) + 1
    starts = np.concatenate(([0], change))
    ends = np.concatenate((change, [n])) - 1
    keep = sounding[starts]
    return starts[keep], ends[keep]


def get_midi_events(rake_mask, f0, voiced_flag, active_probs, rms, sr, hop_length, confidence_threshold,
                    **kwargs):
    noise_gate_db = kwargs.get("noise_gate_db", -40)
    sustain_ms = kwargs.get("sustain_ms", 50)
    min_note_duration_ms = kwargs.get("min_note_duration_ms", 50)

    # The reference tries librosa.util.softmask(f0, ..., margin=0.5) here; that function has no
    # `margin` parameter, the call raises, and the raw track is used (midi_logic.py:41-49).
    f0 = np.asarray(f0)
    rms_db = amplitude_to_db_max(rms)
    min_frames = int((min_note_duration_ms / 1000.0) * sr / hop_length)
    sustain_frames = int((sustain_ms / 1000.0) * sr / hop_length)

    sounding = (np.asarray(voiced_flag, bool) & ~(rms_db < noise_gate_db) & (f0 > 0)
                & ~np.asarray(rake_mask, bool))
    pitch = np.zeros(len(f0), np.int64)
    if sounding.any():
        pitch[sounding] = np.rint(hz_to_midi(f0[sounding])).astype(np.int64)
    starts, ends = _note_runs(sounding, pitch)
    if len(starts) == 0:
        return []

    events = []
    for s, e in zip(starts.tolist(), ends.tolist()):
        energy, conf = rms_db[s], active_probs[s]
        technique, slope = detect_articulations(f0, s, e, sr, hop_length)
        events.append({
            "note": int(pitch[s]), "start": s, "end": e, "confidence": conf,
            "velocity": int(np.clip((energy + 80) * 1.5, 0, 127)),
            "track": "main" if conf >= confidence_threshold else "safe",
            "rms_energy": energy, "technique": technique, "slope": slope})

    events = [ev for ev in events if ev["end"] - ev["start"] >= min_frames]

    if len(events) > 1:          # join same-pitch neighbours across short gaps (no technique only)
        out = [events[0]]
        for ev in events[1:]:
            head = out[-1]
            if (ev["note"] == head["note"] and ev["start"] - head["end"] <= sustain_frames
                    and not head.get("technique")):
                head["end"] = ev["end"]
            else:
                out.append(ev)
        events = out

    frame_ms = (hop_length / sr) * 1000
    for prev, cur in zip(events, events[1:]):
        if (cur["start"] - prev["end"]) * frame_ms < 30:
            interval = cur["note"] - prev["note"]
            softer = (cur["velocity"] / max(prev["velocity"], 1) < 0.7
                      or cur.get("rms_energy", 0) / max(prev.get("rms_energy", 1), -80) < 0.8)
            if softer and 0 < interval <= 2:
                cur["technique"], cur["slope"] = "hammer_on", 0.0
            elif softer and -2 <= interval < 0:
                cur["technique"], cur["slope"] = "pull_off", 0.0
    return events
