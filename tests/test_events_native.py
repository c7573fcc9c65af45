"""the batched C++ event extraction and SMF writer (aegis_extract_events / aegis_render_smf,
host code of libaegis_hip.so -- no GPU needed) against the events the reference's own midi_logic.py produced
(tests/golden/v1_events_golden.*), against the per-clip NumPy path, and byte for byte against the SMF writers."""
import json
import os

import numpy as np
import pytest

from oracle import smf as osmf
from spectrogram_midi_amd import events_native as en, midi_logic, smf
from test_golden_v1_events import KEYS, KW, assert_same

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    arrays = np.load(os.path.join(HERE, "golden", "v1_events_golden.npz"))
    with open(os.path.join(HERE, "golden", "v1_events_golden.json")) as f:
        return arrays, json.load(f)


def native(raw, kw, **extra):
    rest = {k: v for k, v in kw.items() if k != "confidence_threshold"}
    return en.get_midi_events(rake_mask=raw["rake_mask"], f0=raw["f0"], voiced_flag=raw["voiced_flag"],
                              active_probs=raw["voiced_probs"], rms=raw["rms"], sr=44100, hop_length=512,
                              confidence_threshold=kw.get("confidence_threshold", 0.70), **rest, **extra)


def test_native_events_equal_the_reference(golden):
    arrays, meta = golden
    for clip, by_kw in meta["events"].items():
        raw = {k: arrays[f"{clip}/{k}"] for k in KEYS}
        for tag, want in by_kw.items():
            assert_same(native(raw, KW[tag]), want, ("native", clip, tag))
    seen = set()
    for i, case in enumerate(meta["fuzz"]):
        raw = {k: arrays[f"fuzz{i}/{k}"] for k in KEYS}
        ev = native(raw, case["kw"])
        assert_same(ev, case["events"], ("native", "fuzz", i))
        seen |= {e["technique"] for e in ev}
    assert seen == {None, "vibrato", "bend", "slide", "hammer_on", "pull_off"}


def test_event_field_types_follow_the_reference(golden):
    arrays, meta = golden
    raw = {k: arrays[f"notes/{k}"] for k in KEYS}
    a, b = native(raw, {})[0], midi_logic.get_midi_events(rake_mask=raw["rake_mask"], f0=raw["f0"], voiced_flag=raw["voiced_flag"],
                                                          active_probs=raw["voiced_probs"], rms=raw["rms"], sr=44100,
                                                          hop_length=512, confidence_threshold=0.7)[0]
    assert list(a) == list(b)
    for k in a:
        assert type(a[k]) is type(b[k]), (k, type(a[k]), type(b[k]))


def test_ragged_batch_equals_clip_by_clip(golden):
    arrays, meta = golden
    names = ["guitar", "poly", "notes", "scale", "pitched_start"] + [f"fuzz{i}" for i in range(12)]
    raws = [{k: arrays[f"{n}/{k}"] for k in KEYS} for n in names]
    raws.insert(3, {"rake_mask": np.zeros(0, bool), "f0": np.zeros(0), "voiced_flag": np.zeros(0, bool),
                    "voiced_probs": np.zeros(0), "rms": np.zeros(0, np.float32)})       # an empty clip in the middle
    off = np.concatenate([[0], np.cumsum([len(r["f0"]) for r in raws])])
    cat = {k: np.concatenate([r[k] for r in raws]) for k in KEYS}
    # amplitude_to_db of the whole batch at once == clip by clip, bit for bit
    want_db = np.concatenate([midi_logic.amplitude_to_db_max(r["rms"]) if len(r["rms"]) else np.zeros(0, np.float32) for r in raws])
    got_db = en.batch_rms_db(cat["rms"], off)
    assert got_db.dtype == np.float32 and np.array_equal(got_db, want_db)
    for kw in ({}, {"min_note_duration_ms": 100, "sustain_ms": 200}, {"noise_gate_db": -20}):
        per, blobs = en.extract_batch(off, cat["rake_mask"], cat["f0"], cat["voiced_flag"], cat["voiced_probs"], cat["rms"],
                                      44100, 512, 0.7, want_midi=True, midi_program=30, **kw)
        assert len(per) == len(raws) == len(blobs)
        for r, ev, blob in zip(raws, per, blobs):
            ref = midi_logic.get_midi_events(rake_mask=r["rake_mask"], f0=r["f0"], voiced_flag=r["voiced_flag"],
                                             active_probs=r["voiced_probs"], rms=r["rms"], sr=44100, hop_length=512,
                                             confidence_threshold=0.7, **kw) if len(r["f0"]) else []
            assert_same(ev, ref, "batch")
            assert blob == smf.render(ref, 44100, 512, midi_program=30)
        ev_arr, ev_off = en.extract_batch(off, cat["rake_mask"], cat["f0"], cat["voiced_flag"], cat["voiced_probs"], cat["rms"],
                                          44100, 512, 0.7, packed=True, **kw)
        assert list(np.diff(ev_off)) == [len(e) for e in per] and np.array_equal(ev_arr, en.pack(per))


def test_native_smf_is_byte_identical(golden):
    arrays, meta = golden
    n = 0
    for i, case in enumerate(meta["fuzz"]):
        ev = case["events"]
        for kw in ({}, {"midi_program": 30, "vibrato_rate": 6.5, "vibrato_depth": 0.45}):
            blob = en.render_smf(ev, 44100, 512, **kw)
            assert blob == smf.render(ev, 44100, 512, **kw) == osmf.write_smf(ev, 44100, 512, **kw), i
        n += len(ev)
    assert n > 500
    assert en.render_smf([], 22050, 256) == smf.render([], 22050, 256)
    with pytest.raises(ValueError):
        en.render_smf([{"note": 60, "start": 0, "end": 50, "velocity": 90, "track": "main", "technique": "vibrato",
                        "slope": 0.0, "confidence": 0.9, "rms_energy": np.float32(-3)}], 44100, 512, vibrato_depth=1.5)


def test_decisions_next_to_a_threshold_go_through_the_reference_arithmetic():
    """A pitch ramp of exactly 0.05 semitones per frame sits ON the bend threshold: the native path flags the clip and
    the result is the NumPy path's (np.polyfit decides, as in the reference)."""
    n = 40
    semis = 60.0 + 0.05 * np.arange(n)
    f0 = 440.0 * 2 ** ((semis - 69) / 12)
    raw = dict(rake_mask=np.zeros(n, bool), f0=f0, voiced_flag=np.ones(n, bool), active_probs=np.full(n, 0.9),
               rms=np.full(n, 0.1, np.float32))
    ref = midi_logic.get_midi_events(sr=44100, hop_length=512, confidence_threshold=0.7, **raw)
    got = en.get_midi_events(sr=44100, hop_length=512, confidence_threshold=0.7, **raw)
    assert_same(got, ref, "risky")
    assert len(ref) >= 1


def test_table_lookup_of_the_pitch_grid_gives_the_same_events(golden):
    """With the analysis's decoded bins (aegis_outputs.pitch_bin) hz_to_midi(f0) is a 441-entry table: same events, same bytes."""
    from oracle import pyin as opyin
    arrays, meta = golden
    freqs = opyin.PyinParams().freqs
    names = ["guitar", "poly", "notes", "scale", "pitched_start"]
    raws = [{k: arrays[f"{n}/{k}"] for k in KEYS} for n in names]
    off = np.concatenate([[0], np.cumsum([len(r["f0"]) for r in raws])])
    cat = {k: np.concatenate([r[k] for r in raws]) for k in KEYS}
    bins = np.where(cat["voiced_flag"], np.searchsorted(freqs, cat["f0"]).clip(0, 440), -1).astype(np.int16)
    assert np.array_equal(np.where(bins >= 0, freqs[bins.clip(0)], 0.0), cat["f0"])          # f0 sits on the grid
    a = en.extract_batch(off, cat["rake_mask"], cat["f0"], cat["voiced_flag"], cat["voiced_probs"], cat["rms"], 44100, 512, 0.7, want_midi=True)
    b = en.extract_batch(off, cat["rake_mask"], cat["f0"], cat["voiced_flag"], cat["voiced_probs"], cat["rms"], 44100, 512, 0.7, want_midi=True,
                         pitch_bin=bins, freqs=freqs)
    assert a[1] == b[1] and sum(len(e) for e in a[0]) > 40
    for x, y in zip(a[0], b[0]):
        assert_same(x, y, "grid")


def test_event_dicts_built_in_c_are_the_comprehensions(golden, monkeypatch):
    """csrc/pyevents.c (_aegis_pyevents.event_dicts) against the Python comprehension it replaces in extract_batch: equal
    dicts, keys in the same order, values of the same types (np.float64 confidence, np.float32 rms_energy, int, str / None)."""
    if en._pyevents is None:
        pytest.skip("_aegis_pyevents.so not built (csrc/Makefile builds it when Python.h is present)")
    arrays, meta = golden
    clips = list(meta["events"])
    raws = [{k: arrays[f"{c}/{k}"] for k in KEYS} for c in clips] + [{k: arrays[f"fuzz{i}/{k}"] for k in KEYS} for i in range(len(meta["fuzz"]))]
    n = [min(len(r["rake_mask"]), len(r["f0"]), len(r["rms"])) for r in raws]
    off = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
    cat = {k: np.concatenate([r[k][:m] for r, m in zip(raws, n)]) for k in KEYS}
    run = lambda: en.extract_batch(off, cat["rake_mask"], cat["f0"], cat["voiced_flag"], cat["voiced_probs"], cat["rms"], 44100, 512, 0.70)
    fast = run()
    monkeypatch.setattr(en, "_pyevents", None)
    slow = run()
    assert sum(len(c) for c in fast) > 100 and fast == slow
    for a, b in zip(fast, slow):
        for x, y in zip(a, b):
            assert list(x) == list(y) and [type(v) for v in x.values()] == [type(v) for v in y.values()]
    techs = {e["technique"] for c in fast for e in c}
    assert None in techs and len(techs) >= 4
