"""Host-side scalar conversions the event logic needs (the librosa helpers the reference calls
in aegis_engine_core/midi_logic.py:17,51,69 and aegis_engine.py:63): note names, Hz<->MIDI,
amplitude->dB.  Array arithmetic only; the per-frame DSP lives in the HIP kernels."""
import numpy as np

_PITCH_CLASS = dict(C=0, D=2, E=4, F=5, G=7, A=9, B=11)


def note_to_hz(name):
    """'E2' -> 82.4068892282175, 'C6' -> 1046.5022612023945 (equal temperament, A4 = 440)."""
    pc, i = _PITCH_CLASS[name[0].upper()], 1
    while i < len(name) and name[i] in "#b":
        pc += 1 if name[i] == "#" else -1
        i += 1
    midi = 12 * (int(name[i:] or 0) + 1) + pc
    return float(440.0 * 2.0 ** ((midi - 69.0) / 12.0))


def hz_to_midi(freqs):
    return 12 * (np.log2(np.asanyarray(freqs)) - np.log2(440.0)) + 69


def amplitude_to_db_max(x, amin=1e-5, top_db=80.0):
    """amplitude_to_db(x, ref=np.max): 10*log10(max(amin^2, x^2)) - 10*log10(max(amin^2, max|x|^2)),
    floored at (peak - top_db).  Keeps the dtype of x (float32 for the engine's rms)."""
    mag = np.abs(np.asarray(x))
    peak = np.max(mag)
    sq = np.square(mag)
    db = 10.0 * np.log10(np.maximum(amin ** 2, sq))
    db -= 10.0 * np.log10(np.maximum(amin ** 2, peak ** 2))
    return np.maximum(db, db.max() - top_db)
