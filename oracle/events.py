"""Oracle: frame arrays -> note events (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates `/root/reference/aegis_engine_core/midi_logic.py:6-30`
(`detect_articulations`) and `:32-148` (`get_midi_events`) including the quirks
SURVEY.md section 8a lists (Q1: the softmask call always raises, so the raw f0 is
used; Q5-Q7).  librosa helpers come from oracle.dsp.  PINNED by tests/golden/v1_events_golden.json (events produced by the reference module itself).
"""
import numpy as np

from .dsp import amplitude_to_db, hz_to_midi


def articulation_of(f0, first, last):
    """midi_logic.py:6-30 -> (technique | None, slope)."""
    if last <= first:
        return None, 0.0
    seg = f0[first : last + 1]
    seg = seg[seg > 0]
    if len(seg) < 3:
        return None, 0.0
    semis = hz_to_midi(seg)
    x = np.arange(len(semis))
    slope = np.polyfit(x, semis, 1)[0]
    resid = semis - np.polyval(np.polyfit(x, semis, 1), x)
    if np.max(resid) - np.min(resid) > 0.3:
        return "vibrato", slope
    if slope > 0.05:
        return "bend", slope
    if abs(slope) > 0.02:
        return "slide", slope
    return None, 0.0


def _finish(ev, f0):
    ev["technique"], ev["slope"] = articulation_of(f0, ev["start"], ev["end"])
    return ev


def get_midi_events(rake_mask, f0, voiced_flag, active_probs, rms, sr, hop_length,
                    confidence_threshold, **kwargs):
    gate_db = kwargs.get("noise_gate_db", -40)
    sustain_ms = kwargs.get("sustain_ms", 50)
    min_ms = kwargs.get("min_note_duration_ms", 50)

    # midi_logic.py:41-49: librosa.util.softmask has no `margin` argument, the
    # call raises TypeError, and the reference falls through to the raw track.
    track = f0
    level_db = amplitude_to_db(rms)
    min_frames = int((min_ms / 1000.0) * sr / hop_length)
    sustain_frames = int((sustain_ms / 1000.0) * sr / hop_length)

    notes, cur = [], None
    for t in range(len(track)):
        hz, level = track[t], level_db[t]
        sounding = bool(voiced_flag[t]) and not (level < gate_db)
        if sounding and hz > 0 and not rake_mask[t]:
            pitch = int(round(hz_to_midi(hz)))
            if cur is not None and cur["note"] == pitch:
                cur["end"] = t
                continue
            if cur is not None:
                notes.append(_finish(cur, track))
            conf = active_probs[t]
            cur = {"note": pitch, "start": t, "end": t, "confidence": conf,
                   "velocity": int(np.clip((level + 80) * 1.5, 0, 127)),
                   "track": "main" if conf >= confidence_threshold else "safe",
                   "rms_energy": level}
        elif cur is not None:
            notes.append(_finish(cur, track))
            cur = None
    if cur is not None:
        notes.append(_finish(cur, track))
    if not notes:
        return []

    notes = [n for n in notes if (n["end"] - n["start"]) >= min_frames]        # :109

    if len(notes) > 1:                                                         # :112-124
        joined, head = [], notes[0]
        for nxt in notes[1:]:
            if (nxt["note"] == head["note"] and nxt["start"] - head["end"] <= sustain_frames
                    and not head.get("technique")):
                head["end"] = nxt["end"]
            else:
                joined.append(head)
                head = nxt
        joined.append(head)
        notes = joined

    for a, b in zip(notes[:-1], notes[1:]):                                    # :127-146
        gap_ms = (b["start"] - a["end"]) * (hop_length / sr) * 1000
        if not gap_ms < 30:
            continue
        step = b["note"] - a["note"]
        vel_ratio = b["velocity"] / max(a["velocity"], 1)
        level_ratio = b.get("rms_energy", 0) / max(a.get("rms_energy", 1), -80)
        soft = vel_ratio < 0.7 or level_ratio < 0.8
        if 0 < step <= 2 and soft:
            b["technique"], b["slope"] = "hammer_on", 0.0
        elif -2 <= step < 0 and soft:
            b["technique"], b["slope"] = "pull_off", 0.0
    return notes
