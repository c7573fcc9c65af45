"""v2 event extraction: the reference's `get_midi_events_financial`, `detect_articulations_financial` and
`adaptive_confidence_threshold` (/root/reference/aegis_engine_core_v2/midi_logic_financial.py:16-388) over
the GPU trend filters (financial.py) and the harmonic analysis (harmonic.py).  Same arguments, event schema
(`financial_artic`, `financial_slide`, `technique`, `harmonic_valid`, `key_info` on the first event) and quirks:
the last open note of the legacy path carries no 'technique' key, and an all-out-of-scale result raises
IndexError exactly where the reference does."""
import numpy as np

from .convert import amplitude_to_db_max, hz_to_midi
from .financial import FinancialPitchAnalyzer
from .harmonic import HarmonicAnalyzer


def detect_articulations_financial(f0, start, end, analyzer):
    """Dominant Bollinger / MACD label over frames start..end (midi_logic_financial.py:16-72)."""
    if end <= start:
        return None
    seg = f0[start:end + 1]
    seg = seg[~np.isnan(seg)]
    if len(seg) < 3:
        return None
    labels = analyzer.detect_articulation_bollinger(seg, window=min(5, len(seg)), sensitivity=1.5)
    slides = analyzer.detect_slides_macd(seg, threshold=0.3)
    counts = {}
    for a in labels:
        if a and a != "normal":
            counts[a] = counts.get(a, 0) + 1
    n_slide = sum(1 for s in slides if s and s != "normal")
    if n_slide >= 2:
        counts["slide"] = n_slide
    if not counts:
        return None
    name, hits = max(counts.items(), key=lambda kv: kv[1])
    return name if hits / len(labels) >= 0.3 else None


def adaptive_confidence_threshold(confidence_values, method="bollinger"):
    """mean - std (or the 30th percentile) of the positive confidences, clipped to [0.3, 0.8]."""
    pos = confidence_values[confidence_values > 0]
    if len(pos) == 0:
        return 0.5
    if method == "bollinger":
        return np.clip(np.mean(pos) - np.std(pos), 0.3, 0.8)
    if method == "percentile":
        return np.clip(np.percentile(pos, 30), 0.3, 0.8)
    return 0.5


def get_midi_events_financial(rake_mask, f0, voiced_flag, active_probs, rms, sr, hop_length,
                              confidence_threshold=None, verbose=False, **kwargs):
    noise_gate_db = kwargs.get("noise_gate_db", -40)
    sustain_ms = kwargs.get("sustain_ms", 50)
    min_note_duration_ms = kwargs.get("min_note_duration_ms", 50)
    use_financial = kwargs.get("use_financial", True)
    say = print if verbose else (lambda *a, **k: None)
    f0 = np.asarray(f0, dtype=np.float64)
    voiced_flag = np.asarray(voiced_flag, dtype=bool)
    n = len(f0)

    analyzer = FinancialPitchAnalyzer(sr=sr, hop_length=hop_length)
    if use_financial:
        analysis = analyzer.analyze_pitch_financial(np.where(voiced_flag, f0, np.nan), voiced_flag)
        track = analysis["trend"]
        articulations, slides = analysis["articulations"], analysis["slides"]
        combined = active_probs * 0.5 + analysis["confidence"] * 0.5
        if confidence_threshold is None:
            confidence_threshold = adaptive_confidence_threshold(combined, method="bollinger")
            say(f"[Financial] auto threshold: {confidence_threshold:.3f}")
    else:
        # librosa.util.softmask(..., margin=0.5) raises TypeError in the reference -> raw track
        track = f0
        combined = active_probs
        articulations, slides = [None] * n, [None] * n
        if confidence_threshold is None:
            confidence_threshold = 0.7

    level_db = amplitude_to_db_max(rms)
    min_frames = int((min_note_duration_ms / 1000.0) * sr / hop_length)
    sustain_frames = int((sustain_ms / 1000.0) * sr / hop_length)

    with np.errstate(invalid="ignore"):
        sounding = (voiced_flag[:len(track)] & ~np.isnan(track) & ~(level_db[:len(track)] < noise_gate_db)
                    & (track > 0) & ~np.asarray(rake_mask[:len(track)], dtype=bool))
    pitch = np.zeros(len(track), np.int64)
    if sounding.any():
        pitch[sounding] = np.rint(hz_to_midi(track[sounding])).astype(np.int64)

    def close(ev, is_last):
        if use_financial:
            ev["technique"] = ev.get("financial_artic")
        elif not is_last:        # the reference forgets the technique of a note still open at the end
            ev["technique"] = detect_articulations_financial(track, ev["start"], ev["end"], analyzer)
        return ev

    events, cur = [], None
    for t in range(len(track)):
        if sounding[t]:
            artic = articulations[t] if use_financial else None
            if cur is not None and cur["note"] == pitch[t]:
                cur["end"] = t
                if artic and artic != "normal":
                    cur["financial_artic"] = artic
                continue
            if cur is not None:
                events.append(close(cur, False))
            level, conf = level_db[t], combined[t]
            cur = {"note": int(pitch[t]), "start": t, "end": t, "confidence": conf,
                   "velocity": int(np.clip((level + 80) * 1.5, 0, 127)),
                   "track": "main" if conf >= confidence_threshold else "safe",
                   "financial_artic": artic, "financial_slide": slides[t] if use_financial else None}
        elif cur is not None:
            events.append(close(cur, False))
            cur = None
    if cur is not None:
        events.append(close(cur, True))
    if not events:
        return []

    events = [e for e in events if e["end"] - e["start"] >= min_frames]
    if len(events) > 1:
        out = [events[0]]
        for e in events[1:]:
            head = out[-1]
            if e["note"] == head["note"] and e["start"] - head["end"] <= sustain_frames and not head.get("technique"):
                head["end"] = e["end"]
            else:
                out.append(e)
        events = out

    if use_financial and len(events) > 10:
        events = analyzer.filter_ghost_notes_rsi(events, rsi_threshold=70)

    if use_financial and kwargs.get("use_harmonic_filter", True) and len(events) > 5:
        hz = HarmonicAnalyzer()
        frame_ms = (hop_length / sr) * 1000
        notes = np.array([e["note"] for e in events])
        confs = np.array([e["confidence"] for e in events])
        key_info = hz.detect_key(notes)
        say(f"[Harmonic] key: {key_info['key']} {key_info['mode']} ({key_info['confidence']:.2f})")
        _, kept_conf, outside = hz.filter_out_of_scale_notes(notes, confs, key_info,
                                                             tolerance=kwargs.get("harmonic_tolerance", 1))
        if np.sum(outside) > 0:
            kept = []
            for e, bad in zip(events, outside):
                e["harmonic_valid"] = not bad
                if not bad:
                    e["confidence"] = kept_conf[len(kept)]
                    kept.append(e)
            if kept:
                adjusted = hz.adaptive_filter_by_context(
                    np.array([e["note"] for e in kept]), np.array([e["start"] * frame_ms for e in kept]),
                    np.array([e["confidence"] for e in kept]), key_info)
                for e, c in zip(kept, adjusted):
                    e["confidence"] = c
                    e["track"] = "main" if c >= confidence_threshold else "safe"
            events = kept
            events[0]["key_info"] = key_info        # IndexError when everything was out of scale (Q11)
    say(f"[Financial] events: {len(events)}")
    return events
