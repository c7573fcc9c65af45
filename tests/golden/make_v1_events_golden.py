"""Generates tests/golden/v1_events_golden.{npz,json} by running the REFERENCE's own
aegis_engine_core/midi_logic.py (get_midi_events :32-148, detect_articulations :6-30) on frame arrays computed by
the CPU oracle at 44.1 kHz / hop 512.  Build container only (the reference does not travel).

midi_logic.py imports librosa and mido at module level (both absent).  A stub `librosa` supplies the three helpers
it calls -- hz_to_midi, amplitude_to_db (oracle.dsp one-liners) and util.softmask with librosa's real signature, so
the reference's `margin=0.5` call raises TypeError exactly as it does against librosa -- and a stub
`mido` supplies the unused `Message` name.  Every line of event logic executed is the reference's.

    python tests/golden/make_v1_events_golden.py
"""
import contextlib
import importlib.util
import io
import json
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import dsp, engine as oengine   # noqa: E402
from tools import signals   # noqa: E402

REF = "/root/reference/aegis_engine_core/midi_logic.py"
HERE = os.path.dirname(os.path.abspath(__file__))

# the four keyword sets of tests/test_host_logic.py (engine-level keys are stripped by extract_events,
# aegis_engine.py:89-91, before get_midi_events sees them)
KW = {"default": {}, "long_notes": {"min_note_duration_ms": 100, "sustain_ms": 200},
      "gated": {"noise_gate_db": -20, "confidence_threshold": 0.3},
      "program": {"midi_program": 30}}


def clips():
    return {"guitar": signals.guitar_test_track(), "notes": signals.guitar_clip(6.0, seed=11),
            "scale": signals.c_major_scale(44100), "poly": signals.polyphonic_clip(8.0, seed=5),
            # pitched from sample 0: the one place pYIN's initial distribution shows.  Default = librosa's unvoiced
            # start; the "_uniform" entry is the same audio under the alternative start (oracle p_init="uniform")
            "pitched_start": signals.pitched_start_clip(), "pitched_start_uniform": signals.pitched_start_clip()}


def p_init_of(name):
    return "uniform" if name.endswith("_uniform") else "unvoiced"


def fuzz_cases(n_cases=40, seed=20260220):
    """Seeded synthetic frame arrays (no audio) that reach the branches clips rarely do: pitch wobble around the
    vibrato / bend / slide thresholds, short gaps (merge, hammer_on / pull_off), rake flags, level steps."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_cases):
        n = int(rng.integers(1, 400))
        midi = np.repeat(rng.integers(40, 84, n // 7 + 1), 7)[:n] + rng.normal(0, rng.choice([0.0, 0.05, 0.3]), n)
        midi = midi + np.cumsum(rng.choice([0.0, 0.03, -0.03, 0.08], n)) * (rng.random() < 0.5)
        f0 = 440.0 * 2 ** ((midi - 69) / 12)
        voiced = rng.random(n) < rng.choice([0.3, 0.8, 1.0])
        raw = {"f0": np.where(voiced, f0, 0.0), "voiced_flag": voiced, "voiced_probs": rng.random(n),
               "rake_mask": rng.random(n) < 0.05, "rms": (rng.random(n) ** 3).astype(np.float32) + np.float32(1e-7)}
        kw = {"noise_gate_db": float(rng.choice([-40, -20, -60])), "sustain_ms": float(rng.choice([0, 50, 200])),
              "min_note_duration_ms": float(rng.choice([0, 50, 100])), "confidence_threshold": float(rng.choice([0.3, 0.7]))}
        out.append((raw, kw))
    return out


def load_reference():
    lib = types.ModuleType("librosa")
    lib.hz_to_midi = dsp.hz_to_midi
    lib.amplitude_to_db = lambda S, ref=None: dsp.amplitude_to_db(S)
    util = types.ModuleType("librosa.util")

    def softmask(X, X_ref, *, power=1, split_zeros=False):      # librosa's signature: no `margin`
        raise AssertionError("unreachable: the reference always passes margin=")
    util.softmask = softmask
    lib.util = util
    sys.modules["librosa"], sys.modules["librosa.util"] = lib, util
    mido = types.ModuleType("mido")
    mido.Message = object
    sys.modules["mido"] = mido
    spec = importlib.util.spec_from_file_location("ref_midi_logic", REF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def plain(e):
    return {k: (v.item() if isinstance(v, (np.floating, np.integer, np.bool_)) else v) for k, v in e.items()}


def main():
    ML = load_reference()
    arrays, meta = {}, {"semantics": "librosa-0.10-semantics/numpy1-dtypes", "sr": 44100, "hop": 512, "events": {}}
    for name, y in clips().items():
        raw = oengine.audio_to_midi(y, p_init=p_init_of(name))
        for k in ("rake_mask", "f0", "voiced_flag", "voiced_probs", "rms"):
            arrays[f"{name}/{k}"] = raw[k]
        meta["events"][name] = {}
        for tag, kw in KW.items():
            conf = kw.get("confidence_threshold", 0.70)
            rest = {k: v for k, v in kw.items() if k != "confidence_threshold"}
            log = io.StringIO()
            with contextlib.redirect_stdout(log):
                ev = ML.get_midi_events(rake_mask=raw["rake_mask"], f0=raw["f0"], voiced_flag=raw["voiced_flag"],
                                        active_probs=raw["voiced_probs"], rms=raw["rms"], sr=44100, hop_length=512,
                                        confidence_threshold=conf, **rest)
            assert "Pitch Smoothing failed" in log.getvalue()          # Q1: the smoothing branch never runs
            meta["events"][name][tag] = [plain(e) for e in ev]
    meta["fuzz"] = []
    for i, (raw, kw) in enumerate(fuzz_cases()):
        for k, v in raw.items():
            arrays[f"fuzz{i}/{k}"] = v
        rest = {k: v for k, v in kw.items() if k != "confidence_threshold"}
        with contextlib.redirect_stdout(io.StringIO()):
            ev = ML.get_midi_events(rake_mask=raw["rake_mask"], f0=raw["f0"], voiced_flag=raw["voiced_flag"],
                                    active_probs=raw["voiced_probs"], rms=raw["rms"], sr=44100, hop_length=512,
                                    confidence_threshold=kw["confidence_threshold"], **rest)
        meta["fuzz"].append({"kw": kw, "events": [plain(e) for e in ev]})
    np.savez_compressed(os.path.join(HERE, "v1_events_golden.npz"), **arrays)
    with open(os.path.join(HERE, "v1_events_golden.json"), "w") as f:
        json.dump(meta, f, indent=0)
    print({n: {t: len(v) for t, v in d.items()} for n, d in meta["events"].items()})
    tech = {}
    for c in meta["fuzz"]:
        for e in c["events"]:
            tech[e["technique"]] = tech.get(e["technique"], 0) + 1
    print("fuzz events by technique:", tech)


if __name__ == "__main__":
    main()
