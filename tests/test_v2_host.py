"""v2 harmonic analysis (host logic) against goldens produced by the reference's own
harmonic_analysis.py (tests/golden/make_v2_golden.py).  CPU only -- PINNED."""
import json
import os

import numpy as np

from spectrogram_midi_amd.harmonic import HarmonicAnalyzer, apply_harmonic_filter

CASES = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "v2_harmonic_golden.json")))


def test_key_detection_matches_reference():
    hz = HarmonicAnalyzer()
    for c in CASES:
        midi, durs = np.array(c["midi"], dtype=int), np.array(c["durs"])
        k = hz.detect_key(midi)
        assert (k["key"], k["mode"]) == (c["key"]["key"], c["key"]["mode"]), c["name"]
        assert k["confidence"] == c["key"]["confidence"], c["name"]
        kd = hz.detect_key(midi, use_duration=True, durations=durs)
        assert (kd["key"], kd["mode"], kd["confidence"]) == (c["key_dur"]["key"], c["key_dur"]["mode"], c["key_dur"]["confidence"])
    # the reference's tie-break: A natural minor reports as C major
    assert hz.detect_key(np.array([57, 59, 60, 62, 64, 65, 67, 69]))["key"] == "C"


def test_scale_filter_chords_and_context_match_reference():
    hz = HarmonicAnalyzer()
    for c in CASES:
        if not c["midi"]:
            continue
        midi, conf, times = np.array(c["midi"], dtype=int), np.array(c["conf"]), np.array(c["times"])
        for tol in (0, 1, 2):
            fm, fc, mask = hz.filter_out_of_scale_notes(midi, conf, c["key"], tolerance=tol)
            assert mask.astype(int).tolist() == c[f"mask{tol}"], (c["name"], tol)
            np.testing.assert_array_equal(fm, midi[~mask])
        assert hz.analyze_chord_progression(midi, times) == c["chords"], c["name"]
        np.testing.assert_array_equal(hz.adaptive_filter_by_context(midi, times, conf, c["key"]), c["adaptive"])
        r = apply_harmonic_filter(midi, conf, times=times, tolerance=1)
        assert r["filtered_midi"].tolist() == c["apply"]["filtered_midi"]
        np.testing.assert_array_equal(r["filtered_confidence"], c["apply"]["filtered_confidence"])
        assert r["out_of_scale_mask"].astype(int).tolist() == c["apply"]["out_of_scale_mask"]


def test_empty():
    assert HarmonicAnalyzer().detect_key([]) == {"key": "C", "mode": "major", "confidence": 0.0}
    assert HarmonicAnalyzer().analyze_chord_progression(np.array([]), np.array([])) == []
