"""Generates tests/golden/v2_engine_golden.npz by running the REFERENCE's own
aegis_engine_core_v2/guitar_specific.py::apply_guitar_filters and
midi_logic_financial.py::get_midi_events_financial on frame arrays computed by the CPU oracle at the v2
engine's default rate (22 050 Hz, aegis_engine_financial.py:36).  Build container only.

Both modules import librosa / mido at module level (absent).  A stub `librosa` supplying only the four
helpers they call (hz_to_midi, midi_to_hz, amplitude_to_db, util.softmask-that-raises-TypeError like the real
signature does for margin=) and an empty `mido` are registered; the goldens therefore depend on those
restated one-liners (oracle.dsp) but every line of event logic executed is the reference's."""
import contextlib
import importlib.util
import io
import json
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import dsp, engine as oengine, pyin as opyin, rake as orake   # noqa: E402
from tools import signals   # noqa: E402

REF = "/root/reference/aegis_engine_core_v2"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    lib = types.ModuleType("librosa")
    lib.hz_to_midi = dsp.hz_to_midi
    lib.midi_to_hz = dsp.midi_to_hz
    lib.amplitude_to_db = lambda S, ref=None: dsp.amplitude_to_db(S)
    util = types.ModuleType("librosa.util")

    def softmask(X, X_ref, *, power=1, split_zeros=False):      # no `margin`: the reference's call raises
        raise AssertionError("unreachable")
    util.softmask = softmask
    lib.util = util
    sys.modules["librosa"], sys.modules["librosa.util"] = lib, util
    mido = types.ModuleType("mido")
    mido.Message = object
    sys.modules["mido"] = mido
    pkg = types.ModuleType("aegis_engine_core_v2")
    pkg.__path__ = [REF]
    sys.modules["aegis_engine_core_v2"] = pkg
    mods = {}
    for name in ("financial_filters", "financial_analysis", "harmonic_analysis", "guitar_specific", "midi_logic_financial"):
        spec = importlib.util.spec_from_file_location(f"aegis_engine_core_v2.{name}", os.path.join(REF, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods


def main():
    mods = load_reference()
    GS, ML = mods["guitar_specific"], mods["midi_logic_financial"]
    sr, hop = 22050, 512
    clips = {"notes22k": signals.guitar_clip(12.0, sr=sr, seed=21), "fixture22k": signals.guitar_test_track(sr=sr),
             "poly22k": signals.polyphonic_clip(8.0, sr=sr, seed=5), "scale22k": signals.c_major_scale(sr)}
    out, meta = {}, {}
    for name, y in clips.items():
        S_dB = dsp.power_to_db(dsp.melspectrogram(y, sr=sr, hop_length=hop))
        rk = orake.detect_rake_patterns(S_dB, hop, sr, 0.6)
        f0, vf, vp = opyin.pyin(y, sr=sr, hop_length=hop)
        rms = dsp.rms(y, hop_length=hop)
        out[f"{name}/S_dB"], out[f"{name}/rake"], out[f"{name}/f0"] = S_dB, rk, f0
        out[f"{name}/voiced"], out[f"{name}/vprob"], out[f"{name}/rms"] = vf, vp, rms
        g = GS.apply_guitar_filters(f0, vf, S_dB, hop, sr, rk)
        out[f"{name}/g_f0"], out[f"{name}/g_voiced"] = g["f0"], g["voiced"]
        out[f"{name}/g_rake"], out[f"{name}/g_mute"] = g["rake_mask"], g["mute_mask"]
        meta[name] = {"distortion": g["distortion"], "events": {}}
        voiced2 = g["voiced"] & ~g["mute_mask"]
        for tag, kw in (("default", {}), ("fixed_thr", {"confidence_threshold": 0.6, "min_note_duration_ms": 80}),
                        ("no_harm", {"use_harmonic_filter": False, "sustain_ms": 120}),
                        ("legacy", {"use_financial": False})):
            with contextlib.redirect_stdout(io.StringIO()):
                ev = ML.get_midi_events_financial(rake_mask=g["rake_mask"], f0=g["f0"], voiced_flag=voiced2,
                                                  active_probs=vp, rms=rms, sr=sr, hop_length=hop, **kw)
            meta[name]["events"][tag] = [
                {k: (v if not isinstance(v, (np.floating, np.integer)) else v.item()) for k, v in e.items() if k != "key_info"}
                | ({"key_info": {kk: (vv.item() if hasattr(vv, "item") else vv) for kk, vv in e["key_info"].items()}} if "key_info" in e else {})
                for e in ev]
        hp = GS.GuitarSpecificFilters.detect_hammer_on_pull_off(f0)
        meta[name]["hammer"] = [{k: (v.item() if hasattr(v, "item") else v) for k, v in d.items()} for d in hp]
        thr_in = vp * 0.5 + 0.25
        meta[name]["thr_boll"] = float(ML.adaptive_confidence_threshold(thr_in, "bollinger"))
        meta[name]["thr_pct"] = float(ML.adaptive_confidence_threshold(thr_in, "percentile"))
    sub_in = np.array([40, 60, 82, 110, 220, 440, np.nan, 30.0, 41.3], dtype=float)
    s_f0, s_v = GS.GuitarSpecificFilters.filter_subharmonic_noise(sub_in, np.ones_like(sub_in, dtype=bool))
    out["sub/in"], out["sub/f0"], out["sub/voiced"] = sub_in, s_f0, s_v
    np.savez_compressed(os.path.join(HERE, "v2_engine_golden.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "v2_engine_golden.json"), "w"), indent=0, default=float)
    print({k: {t: len(v) for t, v in m["events"].items()} for k, m in meta.items()}, {k: m["distortion"] for k, m in meta.items()})


if __name__ == "__main__":
    main()
