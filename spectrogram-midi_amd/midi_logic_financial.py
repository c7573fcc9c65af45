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


def _note_events_loop(sounding, pitch, level_db, combined, confidence_threshold, track, analyzer, artic=None, slide=None):
    """The reference's frame loop as written (midi_logic_financial.py:205-262).  Runs the legacy path
    (use_financial=False: the technique of a closed note comes from detect_articulations_financial, and a note still
    open at the end gets no 'technique' key at all); with `artic` / `slide` label lists it is the financial path, kept
    as the check of _note_events_from_frames."""
    use_financial = artic is not None

    def close(ev, is_last):
        if use_financial:
            ev["technique"] = ev.get("financial_artic")
        elif not is_last:        # the reference forgets the technique of a note still open at the end
            ev["technique"] = detect_articulations_financial(track, ev["start"], ev["end"], analyzer)
        return ev

    events, cur = [], None
    for t in range(len(sounding)):
        if sounding[t]:
            a = artic[t] if use_financial else None
            if cur is not None and cur["note"] == pitch[t]:
                cur["end"] = t
                if a and a != "normal":
                    cur["financial_artic"] = a
                continue
            if cur is not None:
                events.append(close(cur, False))
            level, conf = level_db[t], combined[t]
            cur = {"note": int(pitch[t]), "start": t, "end": t, "confidence": conf,
                   "velocity": int(np.clip((level + 80) * 1.5, 0, 127)),
                   "track": "main" if conf >= confidence_threshold else "safe",
                   "financial_artic": a, "financial_slide": slide[t] if use_financial else None}
        elif cur is not None:
            events.append(close(cur, False))
            cur = None
    if cur is not None:
        events.append(close(cur, True))
    return events


_ARTIC = (None, "normal", "bend", "vibrato", "noise")          # financial.py codes
_SLIDE = (None, "normal", "slide_up", "slide_down")
_ARTIC_CODE = {a: i for i, a in enumerate(_ARTIC)}
_SLIDE_CODE = {a: i for i, a in enumerate(_SLIDE)}


def _codes(labels, table):
    if isinstance(labels, np.ndarray) and labels.dtype.kind in "iu":
        return labels
    return np.fromiter((table[x] for x in labels), dtype=np.int8, count=len(labels))


def _note_events_from_frames(sounding, pitch, level_db, combined, confidence_threshold, artic, slide):
    """The frame loop of get_midi_events_financial (midi_logic_financial.py:205-262) as array passes, for the financial
    path: notes are the maximal runs of sounding frames with one pitch; a note's fields come from its first frame,
    its `financial_artic` is the LAST label other than None / 'normal' among its later frames (the first frame's label
    when there is none), and `technique` copies it when the note closes."""
    n = len(sounding)
    if n == 0 or not sounding.any():
        return []
    key = np.where(sounding, pitch, np.iinfo(np.int64).min)
    change = np.flatnonzero(key[1:] != key[:-1]) + 1
    starts = np.concatenate(([0], change))
    ends = np.concatenate((change, [n])) - 1
    keep = sounding[starts]
    starts, ends = starts[keep], ends[keep]
    special = np.where(artic >= 2, np.arange(n), -1)
    last_special = np.maximum.accumulate(special)[ends]
    a_code = np.where(last_special > starts, artic[np.maximum(last_special, 0)], artic[starts])
    conf = combined[starts]
    level = level_db[starts]
    velocity = np.clip((level + 80) * 1.5, 0, 127).astype(np.int64)
    return [{"note": nt, "start": s, "end": e, "confidence": c, "velocity": v,
             "track": "main" if c >= confidence_threshold else "safe",
             "financial_artic": _ARTIC[a], "financial_slide": _SLIDE[sl], "technique": _ARTIC[a]}
            for nt, s, e, c, v, a, sl in zip(pitch[starts].tolist(), starts.tolist(), ends.tolist(), conf, velocity.tolist(),
                                              a_code.tolist(), slide[starts].tolist())]


def _financial_notes(rake_mask, f0, voiced_flag, active_probs, rms, sr, hop_length, confidence_threshold, say, analysis,
                     kwargs):
    """get_midi_events_financial up to and including the merge of neighbouring notes (midi_logic_financial.py:117-290).
    -> (events, confidence_threshold, analyzer)"""
    noise_gate_db = kwargs.get("noise_gate_db", -40)
    sustain_ms = kwargs.get("sustain_ms", 50)
    min_note_duration_ms = kwargs.get("min_note_duration_ms", 50)
    use_financial = kwargs.get("use_financial", True)
    f0 = np.asarray(f0, dtype=np.float64)
    voiced_flag = np.asarray(voiced_flag, dtype=bool)
    n = len(f0)

    analyzer = FinancialPitchAnalyzer(sr=sr, hop_length=hop_length)
    if use_financial:
        if analysis is None:
            analysis = analyzer.analyze_pitch_financial_batch([np.where(voiced_flag, f0, np.nan)], labels=False)[0]
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

    if use_financial:
        events = _note_events_from_frames(sounding, pitch, level_db, combined, confidence_threshold,
                                          _codes(articulations, _ARTIC_CODE), _codes(slides, _SLIDE_CODE))
    else:
        events = _note_events_loop(sounding, pitch, level_db, combined, confidence_threshold, track, analyzer)
    if not events:
        return [], confidence_threshold, analyzer

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
    return events, confidence_threshold, analyzer


def _harmonic_stage(events, confidence_threshold, sr, hop_length, say, kwargs):
    """The harmonic filter at the end of get_midi_events_financial (midi_logic_financial.py:330-388)."""
    use_financial = kwargs.get("use_financial", True)
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


def get_midi_events_financial(rake_mask, f0, voiced_flag, active_probs, rms, sr, hop_length,
                              confidence_threshold=None, verbose=False, **kwargs):
    say = print if verbose else (lambda *a, **k: None)
    events, thr, analyzer = _financial_notes(rake_mask, f0, voiced_flag, active_probs, rms, sr, hop_length,
                                             confidence_threshold, say, None, kwargs)
    if not events:
        return []
    if kwargs.get("use_financial", True) and len(events) > 10:
        events = analyzer.filter_ghost_notes_rsi_batch([events], rsi_threshold=70)[0]
    return _harmonic_stage(events, thr, sr, hop_length, say, kwargs)


def get_midi_events_financial_batch(clips, sr, hop_length, confidence_threshold=None, verbose=False, return_exceptions=False,
                                    **kwargs):
    """get_midi_events_financial for several clips (each a dict with rake_mask, f0, voiced_flag, active_probs, rms):
    ONE fused pitch-analysis call over all the pitch tracks (FinancialPitchAnalyzer.analyze_pitch_financial_batch) and
    ONE RSI call over all the ghost-note density tracks instead of eight library calls per clip; the per-note logic is
    the single-clip function's.  -> list of event lists.

    A pitch track shorter than the Bollinger window (10 frames: an empty or sub-0.25 s file) makes the single-clip
    function raise IndexError, as the reference does (financial_analysis.py:113-146 through np.convolve).  Such clips stay out
    of the fused calls, so one tiny file in a folder cannot lose the other clips' results: with return_exceptions=True
    their element is the exception the single-clip call would have raised, otherwise it is raised once every other clip
    has been processed -- with the complete result list attached as `.results`."""
    say = print if verbose else (lambda *a, **k: None)
    use_financial = kwargs.get("use_financial", True)
    errors = {}
    if use_financial:
        for i, c in enumerate(clips):
            if len(c["f0"]) < 10:
                errors[i] = IndexError(f"clip {i}: series of {len(c['f0'])} samples is shorter than the window 10")
    live = [i for i in range(len(clips)) if i not in errors]
    analyses = {i: None for i in live}
    if use_financial and live:
        an = FinancialPitchAnalyzer(sr=sr, hop_length=hop_length)
        tracks = [np.where(np.asarray(clips[i]["voiced_flag"], bool), np.asarray(clips[i]["f0"], np.float64), np.nan) for i in live]
        analyses = dict(zip(live, an.analyze_pitch_financial_batch(tracks, labels=False)))
    staged = {i: _financial_notes(clips[i]["rake_mask"], clips[i]["f0"], clips[i]["voiced_flag"], clips[i]["active_probs"],
                                  clips[i]["rms"], sr, hop_length, confidence_threshold, say, analyses[i], kwargs) for i in live}
    lists = {i: staged[i][0] for i in live}
    if use_financial:
        need = [i for i in live if len(lists[i]) > 10]
        if need:
            kept = FinancialPitchAnalyzer(sr=sr, hop_length=hop_length).filter_ghost_notes_rsi_batch([lists[i] for i in need])
            for i, ev in zip(need, kept):
                lists[i] = ev
    out = [errors[i] if i in errors else (_harmonic_stage(lists[i], staged[i][1], sr, hop_length, say, kwargs) if lists[i] else [])
           for i in range(len(clips))]
    if errors and not return_exceptions:
        first = errors[min(errors)]
        first.results = out
        raise first
    return out
