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


_TECH = (None, "vibrato", "bend", "slide")


def _articulations_batch(semitones, starts, ends, f0=None, sr=None, hop_length=None):
    """detect_articulations for every run at once.  Inside a run every frame is sounding (f0 > 0), so the fit runs
    over x = 0..n-1 and the least-squares line has the closed form slope = (n Sxy - Sx Sy) / (n Sxx - Sx^2);
    per-run sums, minima and maxima come from ufunc.reduceat over the concatenated run frames (no prefix-sum
    cancellation).  np.polyfit solves the same problem through an SVD: slopes agree to ~1e-13.
    Returns (technique code per run, slope per run); code indexes _TECH."""
    n = ends - starts + 1
    code = np.zeros(len(starts), np.int64)
    slope_out = np.zeros(len(starts))
    fit = np.flatnonzero(n >= 3)
    if len(fit) == 0:
        return code, slope_out
    nf = n[fit]
    seg = np.concatenate(([0], np.cumsum(nf)[:-1]))                    # offset of each run in the compact arrays
    x = np.arange(int(nf.sum())) - np.repeat(seg, nf)                   # 0..n-1 inside every run
    y = semitones[np.repeat(starts[fit], nf) + x]
    nn = nf.astype(np.float64)
    sx = nn * (nn - 1) / 2
    sxx = (nn - 1) * nn * (2 * nn - 1) / 6
    sy = np.add.reduceat(y, seg)
    sxy = np.add.reduceat(x * y, seg)
    slope = (nn * sxy - sx * sy) / (nn * sxx - sx * sx)
    icpt = (sy - slope * sx) / nn
    wobble = y - (np.repeat(slope, nf) * x + np.repeat(icpt, nf))
    spread = np.maximum.reduceat(wobble, seg) - np.minimum.reduceat(wobble, seg)
    c = np.where(spread > 0.3, 1, np.where(slope > 0.05, 2, np.where(np.abs(slope) > 0.02, 3, 0)))
    code[fit] = c
    slope_out[fit] = np.where(c > 0, slope, 0.0)
    if f0 is not None:
        eps = 1e-9
        risky = (np.abs(spread - 0.3) < eps) | (np.abs(slope - 0.05) < eps) | (np.abs(np.abs(slope) - 0.02) < eps)
        for k in fit[risky].tolist():
            tech, sl = detect_articulations(f0, int(starts[k]), int(ends[k]), sr, hop_length)
            code[k], slope_out[k] = _TECH.index(tech), sl
    return code, slope_out


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
    semitones = np.zeros(len(f0))
    if sounding.any():
        semitones[sounding] = hz_to_midi(f0[sounding])
        pitch[sounding] = np.rint(semitones[sounding]).astype(np.int64)
    starts, ends = _note_runs(sounding, pitch)
    # notes shorter than the minimum are dropped before anything looks at their technique (midi_logic.py:109)
    keep = ends - starts >= min_frames
    starts, ends = starts[keep], ends[keep]
    if len(starts) == 0:
        return []

    code, slopes = _articulations_batch(semitones, starts, ends, f0, sr, hop_length)
    energy, conf = rms_db[starts], np.asarray(active_probs)[starts]
    velocity = np.clip((energy + 80) * 1.5, 0, 127).astype(np.int64)
    # confidence / rms_energy stay NumPy scalars (float64 / float32) as in the reference: the hammer-on test below
    # divides two rms_energy values, and that division must stay a float32 one
    events = [{"note": nt, "start": s, "end": e, "confidence": conf[i], "velocity": vel,
               "track": "main" if conf[i] >= confidence_threshold else "safe",
               "rms_energy": energy[i], "technique": _TECH[cd], "slope": sl}
              for i, (nt, s, e, vel, cd, sl) in enumerate(zip(pitch[starts].tolist(), starts.tolist(), ends.tolist(),
                                                              velocity.tolist(), code.tolist(), slopes.tolist()))]

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
