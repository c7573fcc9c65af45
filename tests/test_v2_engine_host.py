"""v2 engine host logic (guitar filters, financial event extraction building blocks) against goldens produced
by the reference's own guitar_specific.py (tests/golden/make_v2_engine_golden.py).  CPU only -- PINNED
(with the four librosa one-liners stubbed, see the generator)."""
import json
import os

import numpy as np

from spectrogram_midi_amd.guitar import GuitarSpecificFilters, apply_guitar_filters
from spectrogram_midi_amd.midi_logic_financial import adaptive_confidence_threshold

HERE = os.path.dirname(__file__)
G = np.load(os.path.join(HERE, "golden", "v2_engine_golden.npz"))
META = json.load(open(os.path.join(HERE, "golden", "v2_engine_golden.json")))


def test_guitar_filters_match_reference():
    for name, m in META.items():
        r = apply_guitar_filters(G[f"{name}/f0"], G[f"{name}/voiced"], G[f"{name}/S_dB"], 512, 22050, G[f"{name}/rake"])
        np.testing.assert_array_equal(r["f0"], G[f"{name}/g_f0"])
        np.testing.assert_array_equal(r["voiced"], G[f"{name}/g_voiced"])
        np.testing.assert_array_equal(r["rake_mask"], G[f"{name}/g_rake"])
        np.testing.assert_array_equal(r["mute_mask"], G[f"{name}/g_mute"])
        assert r["distortion"] == m["distortion"]
        hp = GuitarSpecificFilters.detect_hammer_on_pull_off(G[f"{name}/f0"])
        assert [(d["start"], d["end"], d["type"]) for d in hp] == [(d["start"], d["end"], d["type"]) for d in m["hammer"]]
        np.testing.assert_allclose([d["semitones"] for d in hp], [d["semitones"] for d in m["hammer"]], rtol=1e-12)


def test_subharmonic_known_answer():
    f0, v = GuitarSpecificFilters.filter_subharmonic_noise(G["sub/in"], np.ones(len(G["sub/in"]), bool))
    np.testing.assert_array_equal(f0, G["sub/f0"])
    np.testing.assert_array_equal(v, G["sub/voiced"])


def test_adaptive_threshold():
    for name, m in META.items():
        c = G[f"{name}/vprob"] * 0.5 + 0.25
        assert float(adaptive_confidence_threshold(c, "bollinger")) == m["thr_boll"]
        assert float(adaptive_confidence_threshold(c, "percentile")) == m["thr_pct"]
    assert adaptive_confidence_threshold(np.zeros(4)) == 0.5


def test_vectorised_note_frames_equal_the_reference_loop():
    """The financial path's frame loop (midi_logic_financial.py:205-262) as array passes == the loop as written."""
    from spectrogram_midi_amd import midi_logic_financial as m
    rng = np.random.default_rng(1)
    for case in range(200):
        n = int(rng.integers(1, 300))
        sounding = rng.random(n) < rng.choice([0.3, 0.8, 1.0])
        pitch = np.repeat(rng.integers(40, 80, n // 5 + 1), 5)[:n].astype(np.int64)
        level = (rng.random(n) * -60).astype(np.float32)
        comb = rng.random(n)
        art = rng.choice(5, n, p=[.1, .6, .1, .1, .1]).astype(np.int8)
        sl = rng.integers(0, 4, n).astype(np.int8)
        a = m._note_events_from_frames(sounding, pitch, level, comb, 0.5, art, sl)
        b = m._note_events_loop(sounding, pitch, level, comb, 0.5, None, None, artic=[m._ARTIC[c] for c in art],
                                slide=[m._SLIDE[c] for c in sl])
        assert a == b, case
        for x, y in zip(a, b):
            assert [type(x[k]) for k in y] == [type(y[k]) for k in y]
    assert m._note_events_from_frames(np.zeros(0, bool), np.zeros(0, np.int64), np.zeros(0, np.float32), np.zeros(0), 0.5,
                                      np.zeros(0, np.int8), np.zeros(0, np.int8)) == []
