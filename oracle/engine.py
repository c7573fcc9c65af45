"""Oracle: the analyze call end to end (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates `/root/reference/aegis_engine.py:41-75` (`AegisEngine.audio_to_midi`,
from decoded PCM onward), `:77-96` (`extract_events`) and `:183-216`
(`_parallel_pitch_tracking` chunking; the chunks run serially here or in a
process pool -- the arithmetic per chunk is `_pyin_worker`, worker.py:3-15).
"""
import numpy as np

from . import dsp, events, pyin as _pyin, rake, smf

FMIN = dsp.note_to_hz("E2")
FMAX = dsp.note_to_hz("C6")


def load_features(y, sr=44100, hop_length=512, n_fft=2048):
    """aegis_engine.py:25-26 -> S_dB float32[128, F]."""
    S = dsp.melspectrogram(y, sr=sr, n_fft=n_fft, hop_length=hop_length)
    return dsp.power_to_db(S)


def turbo_chunks(n_samples, sr, hop_length, num_cores):
    """aegis_engine.py:192-204 -> list of (start_sample, stop_sample)."""
    total_frames = int(np.ceil(n_samples / hop_length))
    per = total_frames // num_cores
    if per == 0:
        per = total_frames
    spans = []
    for i in range(num_cores):
        sf = i * per
        if sf >= total_frames:
            break
        ef = (i + 1) * per if i < num_cores - 1 else total_frames
        lo, hi = sf * hop_length, min(ef * hop_length, n_samples)
        if hi > lo:
            spans.append((lo, hi))
    return spans


def _chunk_worker(args):
    chunk, sr, hop = args[:3]
    p_init = args[3] if len(args) > 3 else "unvoiced"
    return _pyin.pyin(chunk, sr=sr, hop_length=hop, fmin=FMIN, fmax=FMAX, p_init=p_init)


def parallel_pitch_tracking(y, sr=44100, hop_length=512, num_cores=8, pool=None, p_init="unvoiced"):
    """aegis_engine.py:183-216 with an explicit core count (p_init: oracle.pyin.initial_distribution)."""
    if len(y) / sr < 5.0:
        return _pyin.pyin(y, sr=sr, hop_length=hop_length, fmin=FMIN, fmax=FMAX, p_init=p_init)
    args = [(y[lo:hi], sr, hop_length, p_init) for lo, hi in turbo_chunks(len(y), sr, hop_length, num_cores)]
    results = list(pool.map(_chunk_worker, args)) if pool is not None else [_chunk_worker(a) for a in args]
    f0s, flags, probs = zip(*results)
    return np.concatenate(f0s), np.concatenate(flags), np.concatenate(probs)


def audio_to_midi(y, sr=44100, hop_length=512, n_fft=2048, rake_sensitivity=0.6,
                  turbo_mode=False, num_cores=8, pool=None, p_init="unvoiced"):
    """aegis_engine.py:50-75 from decoded PCM onward."""
    y = np.asarray(y, dtype=np.float32)
    if len(y) == 0:
        return None
    S_dB = load_features(y, sr, hop_length, n_fft)
    rake_mask = rake.detect_rake_patterns(S_dB, hop_length, sr, rake_sensitivity)
    if turbo_mode:
        f0, voiced_flag, voiced_probs = parallel_pitch_tracking(y, sr, hop_length, num_cores, pool, p_init)
    else:
        f0, voiced_flag, voiced_probs = _pyin.pyin(y, sr=sr, hop_length=hop_length, fmin=FMIN, fmax=FMAX, p_init=p_init)
    f0 = np.nan_to_num(f0)
    level = dsp.rms(y, hop_length=hop_length)
    return {"rake_mask": rake_mask, "f0": f0, "voiced_flag": voiced_flag,
            "voiced_probs": voiced_probs, "rms": level, "y": y, "S_dB": S_dB}


_ENGINE_KEYS = ("confidence_threshold", "start_time", "end_time", "turbo_mode",
                "rake_sensitivity", "vibrato_rate", "vibrato_depth")


def extract_events(raw, sr=44100, hop_length=512, want_smf=False, **kwargs):
    """aegis_engine.py:77-181 -> events (and SMF bytes when want_smf)."""
    names = ("rake_mask", "f0", "voiced_flag", "voiced_probs", "rms")
    n = min(len(raw["rake_mask"]), len(raw["f0"]), len(raw["rms"]))
    arrs = {k: raw[k][:n] for k in names}
    passthrough = {k: v for k, v in kwargs.items() if k not in _ENGINE_KEYS}
    evs = events.get_midi_events(
        rake_mask=arrs["rake_mask"], f0=arrs["f0"], voiced_flag=arrs["voiced_flag"],
        active_probs=arrs["voiced_probs"], rms=arrs["rms"], sr=sr, hop_length=hop_length,
        confidence_threshold=kwargs.get("confidence_threshold", 0.70), **passthrough)
    if not want_smf:
        return evs
    blob = smf.write_smf(evs, sr, hop_length, midi_program=kwargs.get("midi_program", 27),
                         vibrato_rate=kwargs.get("vibrato_rate", 5.0),
                         vibrato_depth=kwargs.get("vibrato_depth", 0.3))
    return evs, blob
