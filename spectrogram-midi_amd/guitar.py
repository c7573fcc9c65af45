"""Electric-guitar specific post-filters of the v2 engine: the reference's `GuitarSpecificFilters` and
`apply_guitar_filters` (/root/reference/aegis_engine_core_v2/guitar_specific.py:15-277).  They consume the
dB mel image and frame arrays the kernels produced; the arithmetic is a handful of column means over
[128, F] plus run-length logic, kept on the host next to the event logic (same names, arguments, returns,
including the quirk that negative dB means make `classify_distortion_level` answer 'heavy')."""
import numpy as np

from .convert import hz_to_midi


def _runs(mask):
    """(start, stop) of every run of True that is closed by a False (a run open at the end is not reported)."""
    m = np.asarray(mask, dtype=bool)
    edges = np.diff(np.concatenate(([False], m, [False])).astype(np.int8))
    starts, stops = np.flatnonzero(edges == 1), np.flatnonzero(edges == -1)
    return [(s, e) for s, e in zip(starts, stops) if e < len(m)]


class GuitarSpecificFilters:
    GUITAR_E2_HZ = 440.0 * 2.0 ** ((40 - 69.0) / 12.0)
    GUITAR_E6_HZ = 440.0 * 2.0 ** ((88 - 69.0) / 12.0)

    @staticmethod
    def filter_subharmonic_noise(f0, voiced_flag, fmin_hz=82.4):
        """Frames below E2 are dropped, unless doubling the frequency lands in [fmin, 4*fmin): then the
        octave error is corrected instead (guitar_specific.py:23-58)."""
        f0 = np.asarray(f0, dtype=np.float64)
        out_f0, out_v = f0.copy(), np.asarray(voiced_flag).copy()
        with np.errstate(invalid="ignore"):
            low = f0 < fmin_hz
            doubled = f0 * 2
            fix = low & (doubled >= fmin_hz) & (doubled < fmin_hz * 4)
        out_f0[low], out_v[low] = np.nan, False
        out_f0[fix], out_v[fix] = doubled[fix], True
        return out_f0, out_v

    @staticmethod
    def detect_palm_mute(S_dB, hop_length, sr, duration_ms=50, _means=None):
        """Low-half / high-half mean ratio above 2 for at most duration_ms (guitar_specific.py:60-103).
        _means = (low-half, high-half) column means when the analysis delivered them (aegis_outputs.sdb_col_means)."""
        if _means is None:
            n_mels, F = S_dB.shape
            mid = n_mels // 2
            lo, hi = np.mean(S_dB[:mid, :], axis=0), np.mean(S_dB[mid:, :], axis=0)
        else:
            lo, hi = _means
            F = len(lo)
        ratio = lo / (hi + 1e-6)
        max_frames = int(duration_ms / ((hop_length / sr) * 1000))
        out = np.zeros(F, dtype=bool)
        for s, e in _runs(ratio > 2.0):
            if e - s <= max_frames:
                out[s:e] = True
        return out

    @staticmethod
    def detect_rake_enhanced(S_dB, hop_length, sr, rake_mask_basic, _level=None):
        """Adds frames after a >10 dB jump of the mean level when the next 30 ms fall on average
        (guitar_specific.py:105-141).  _level = np.mean(S_dB, axis=0) when the analysis delivered it."""
        out = np.asarray(rake_mask_basic).copy()
        level = np.mean(S_dB, axis=0) if _level is None else _level
        diff = np.diff(level, prepend=level[0])
        span = int(30 / ((hop_length / sr) * 1000))
        for i in np.flatnonzero(diff[1:] > 10) + 1:
            if i + span < len(diff) and np.mean(diff[i:i + span]) < 0:
                out[i:i + span] = True
        return out

    @staticmethod
    def detect_hammer_on_pull_off(f0, min_semitone_jump=2, max_duration_ms=100):
        """Frame-to-frame jumps of >= min_semitone_jump semitones (guitar_specific.py:143-197)."""
        f0 = np.asarray(f0, dtype=np.float64)
        ok = ~np.isnan(f0)
        if not np.any(ok):
            return []
        midi = np.full_like(f0, np.nan)
        midi[ok] = hz_to_midi(f0[ok])
        found = []
        for i in range(1, len(midi) - 1):
            if np.isnan(midi[i]) or np.isnan(midi[i - 1]):
                continue
            jump = midi[i] - midi[i - 1]
            if abs(jump) >= min_semitone_jump:
                held = 1
                for j in range(i + 1, min(i + 10, len(midi))):
                    if np.isnan(midi[j]) or abs(midi[j] - midi[i]) > 0.5:
                        break
                    held += 1
                found.append({"start": i, "end": i + held, "type": "hammer_on" if jump > 0 else "pull_off",
                              "semitones": abs(jump)})
        return found

    @staticmethod
    def classify_distortion_level(S_dB):
        n_mels = S_dB.shape[0]
        ratio = np.mean(S_dB[int(n_mels * 0.7):, :]) / (np.mean(S_dB) + 1e-6)
        return "heavy" if ratio > 0.4 else ("light" if ratio > 0.25 else "clean")


def apply_guitar_filters(f0, voiced_flag, S_dB, hop_length, sr, rake_mask):
    g = GuitarSpecificFilters
    f0_f, voiced_f = g.filter_subharmonic_noise(f0, voiced_flag, fmin_hz=82.4)
    return {"f0": f0_f, "voiced": voiced_f, "rake_mask": g.detect_rake_enhanced(S_dB, hop_length, sr, rake_mask),
            "mute_mask": g.detect_palm_mute(S_dB, hop_length, sr), "distortion": g.classify_distortion_level(S_dB)}


def apply_guitar_filters_from_means(f0, voiced_flag, col_means, hop_length, sr, rake_mask):
    """apply_guitar_filters on the three column means of the dB image the library computes beside it
    (col_means = (all rows, low half, high half), aegis_outputs.sdb_col_means) instead of the image itself; the
    'distortion' label, which needs the whole image and which no caller of the engine reads, is left out."""
    g = GuitarSpecificFilters
    f0_f, voiced_f = g.filter_subharmonic_noise(f0, voiced_flag, fmin_hz=82.4)
    return {"f0": f0_f, "voiced": voiced_f,
            "rake_mask": g.detect_rake_enhanced(None, hop_length, sr, rake_mask, _level=col_means[0]),
            "mute_mask": g.detect_palm_mute(None, hop_length, sr, _means=(col_means[1], col_means[2]))}
