"""Key / scale / chord-context filtering of note lists: the reference's `HarmonicAnalyzer` and
`apply_harmonic_filter` (/root/reference/aegis_engine_core_v2/harmonic_analysis.py:15-330).  These run on
tens to hundreds of notes per clip (host logic, not a kernel); same names, arguments,
return values and tie-breaking (first root/mode tested wins, so a relative minor reports as its major)."""
from collections import Counter

import numpy as np

CHROMATIC = ["C", "C#", "D", "D#", "E", "F", "F#", "G", "G#", "A", "A#", "B"]
_SCALES = {"major": (0, 2, 4, 5, 7, 9, 11), "minor": (0, 2, 3, 5, 7, 8, 10), "blues": (0, 3, 5, 6, 7, 10)}


class HarmonicAnalyzer:
    CHROMATIC = CHROMATIC
    MAJOR_INTERVALS = list(_SCALES["major"])
    MINOR_INTERVALS = list(_SCALES["minor"])
    BLUES_INTERVALS = list(_SCALES["blues"])
    PENTA_MINOR_INTERVALS = [0, 3, 5, 7, 10]

    @staticmethod
    def midi_to_pitch_class(midi_note):
        return int(midi_note) % 12

    def detect_key(self, midi_notes, use_duration=False, durations=None):
        if len(midi_notes) == 0:
            return {"key": "C", "mode": "major", "confidence": 0.0}
        weights = durations if (use_duration and durations is not None) else np.ones(len(midi_notes))
        hist = np.zeros(12)
        for note, w in zip(midi_notes, weights):
            hist[int(note) % 12] += w
        hist = hist / (np.sum(hist) + 1e-6)
        best = ("C", "major", 0.0)
        for root in range(12):
            for mode in ("major", "minor", "blues"):
                score = 0.0
                for step in _SCALES[mode]:
                    score += hist[(root + step) % 12]
                if score > best[2]:
                    best = (CHROMATIC[root], mode, score)
        return {"key": best[0], "mode": best[1], "confidence": best[2]}

    def get_scale_notes(self, key, mode):
        root = CHROMATIC.index(key)
        return [(root + step) % 12 for step in _SCALES.get(mode, _SCALES["major"])]

    def filter_out_of_scale_notes(self, midi_notes, confidences, key_info, tolerance=1):
        scale = self.get_scale_notes(key_info["key"], key_info["mode"])
        out = np.zeros(len(midi_notes), dtype=bool)
        for i, note in enumerate(midi_notes):
            pc = int(note) % 12
            out[i] = min(min(abs(pc - s), 12 - abs(pc - s)) for s in scale) > tolerance
        return midi_notes[~out], confidences[~out], out

    def analyze_chord_progression(self, midi_notes, times, window_size=2000):
        if len(midi_notes) == 0:
            return []
        chords = []
        for t in range(0, int(np.max(times)), window_size):
            inside = midi_notes[(times >= t) & (times < t + window_size)]
            if len(inside) == 0:
                continue
            pcs = [int(n) % 12 for n in inside]
            root = Counter(pcs).most_common(1)[0][0]
            quality = "major" if (root + 4) % 12 in pcs else ("minor" if (root + 3) % 12 in pcs else "unknown")
            chords.append({"time": t, "chord": CHROMATIC[root], "quality": quality})
        return chords

    def adaptive_filter_by_context(self, midi_notes, times, confidences, key_info):
        chords = self.analyze_chord_progression(midi_notes, times)
        if len(chords) == 0:
            return confidences
        adjusted = confidences.copy()
        scale = self.get_scale_notes(key_info["key"], key_info["mode"])
        for i, (note, time) in enumerate(zip(midi_notes, times)):
            chord = next((c for c in chords if c["time"] <= time < c["time"] + 2000), None)
            if chord is None or chord["quality"] == "unknown":
                continue
            root = CHROMATIC.index(chord["chord"])
            tones = (root, (root + (4 if chord["quality"] == "major" else 3)) % 12, (root + 7) % 12)
            pc = int(note) % 12
            if pc not in tones:
                adjusted[i] *= 0.8 if pc in scale else 0.5
        return adjusted


def apply_harmonic_filter(midi_notes, confidences, times=None, tolerance=1, verbose=False):
    analyzer = HarmonicAnalyzer()
    key_info = analyzer.detect_key(midi_notes)
    if verbose:
        print(f"[Harmonic] key: {key_info['key']} {key_info['mode']} (confidence: {key_info['confidence']:.2f})")
    kept, conf, mask = analyzer.filter_out_of_scale_notes(midi_notes, confidences, key_info, tolerance)
    if times is not None:
        conf = analyzer.adaptive_filter_by_context(kept, times[~mask], conf, key_info)
    return {"key_info": key_info, "filtered_midi": kept, "filtered_confidence": conf, "out_of_scale_mask": mask}
