"""SURVEY 8a row a11 PINNED: note events of the product's midi_logic and of the oracle's restatement against
events produced by the reference's own aegis_engine_core/midi_logic.py (tests/golden/make_v1_events_golden.py runs
it under a stub librosa on the oracle's frame arrays and on seeded synthetic frame arrays).  Integer fields, track,
technique, confidence and level are exact; the articulation slope (np.polyfit's SVD in the reference, closed-form
least squares in the product) to 1e-9."""
import json
import os

import numpy as np
import pytest

from oracle import events as oevents
from spectrogram_midi_amd import midi_logic

HERE = os.path.dirname(os.path.abspath(__file__))
KEYS = ("rake_mask", "f0", "voiced_flag", "voiced_probs", "rms")


@pytest.fixture(scope="module")
def golden():
    arrays = np.load(os.path.join(HERE, "golden", "v1_events_golden.npz"))
    with open(os.path.join(HERE, "golden", "v1_events_golden.json")) as f:
        meta = json.load(f)
    return arrays, meta


def assert_same(events, want, what, slope_tol=1e-9):
    assert len(events) == len(want), what
    for i, (e, g) in enumerate(zip(events, want)):
        assert set(e) == set(g), (what, i)
        for k in g:
            if k == "slope":
                assert abs(e[k] - g[k]) <= slope_tol * max(1.0, abs(g[k])), (what, i, k, e[k], g[k])
            elif isinstance(g[k], float):
                assert e[k] == g[k] or (np.isnan(e[k]) and np.isnan(g[k])) or (np.isinf(g[k]) and e[k] == g[k]), (what, i, k, e[k], g[k])
            else:
                assert e[k] == g[k], (what, i, k, e[k], g[k])


def run(fn, arrays, prefix, kw):
    raw = {k: arrays[f"{prefix}/{k}"] for k in KEYS}
    rest = {k: v for k, v in kw.items() if k != "confidence_threshold"}
    return fn(rake_mask=raw["rake_mask"], f0=raw["f0"], voiced_flag=raw["voiced_flag"], active_probs=raw["voiced_probs"],
              rms=raw["rms"], sr=44100, hop_length=512, confidence_threshold=kw.get("confidence_threshold", 0.70), **rest)


KW = {"default": {}, "long_notes": {"min_note_duration_ms": 100, "sustain_ms": 200},
      "gated": {"noise_gate_db": -20, "confidence_threshold": 0.3}, "program": {"midi_program": 30}}


@pytest.mark.parametrize("impl", ["product", "oracle"])
def test_clip_events_equal_the_reference(golden, impl):
    arrays, meta = golden
    fn = midi_logic.get_midi_events if impl == "product" else oevents.get_midi_events
    assert meta["semantics"] == "librosa-0.10-semantics/numpy1-dtypes"
    for clip, by_kw in meta["events"].items():
        for tag, want in by_kw.items():
            assert_same(run(fn, arrays, clip, KW[tag]), want, (impl, clip, tag))


@pytest.mark.parametrize("impl", ["product", "oracle"])
def test_fuzz_events_equal_the_reference(golden, impl):
    arrays, meta = golden
    fn = midi_logic.get_midi_events if impl == "product" else oevents.get_midi_events
    seen = set()
    for i, case in enumerate(meta["fuzz"]):
        ev = run(fn, arrays, f"fuzz{i}", case["kw"])
        assert_same(ev, case["events"], (impl, "fuzz", i))
        seen |= {e["technique"] for e in ev}
    assert seen == {None, "vibrato", "bend", "slide", "hammer_on", "pull_off"}       # every branch is pinned


@pytest.mark.gpu
def test_gpu_raw_data_gives_the_reference_events(golden):
    """The same clips through the HIP path (C ABI) and the product's event logic: events equal what the reference's
    midi_logic.py produced from the oracle's frame arrays."""
    from tools import signals
    from spectrogram_midi_amd.engine import AegisEngine
    arrays, meta = golden
    clips = {"guitar": signals.guitar_test_track(), "notes": signals.guitar_clip(6.0, seed=11),
             "scale": signals.c_major_scale(44100), "poly": signals.polyphonic_clip(8.0, seed=5),
             "pitched_start": signals.pitched_start_clip()}
    assert set(clips) | {"pitched_start_uniform"} == set(meta["events"])
    # librosa's unvoiced start (the default) and the uniform start differ on a clip pitched from sample 0: frame 0
    assert not arrays["pitched_start/voiced_flag"][0] and arrays["pitched_start_uniform/voiced_flag"][0]
    for mode, names in (("unvoiced", list(clips)), ("uniform", ["pitched_start"])):
        eng = AegisEngine(pyin_init=mode)
        raws = eng.analyze_arrays([clips[n] for n in names])
        for name, raw in zip(names, raws):
            clip = name + ("_uniform" if mode == "uniform" else "")
            for k in ("rake_mask", "voiced_flag", "rms", "voiced_probs", "f0"):
                np.testing.assert_array_equal(raw[k], arrays[f"{clip}/{k}"], err_msg=f"{clip} {k}")
            for tag, want in meta["events"][clip].items():
                assert_same(eng.extract_events(raw, None, **KW[tag]), want, ("gpu", clip, tag))
        eng.close()
