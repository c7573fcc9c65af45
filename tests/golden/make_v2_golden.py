"""Generates tests/golden/v2_trend_golden.npz and v2_harmonic_golden.json by running the REFERENCE's own
v2 modules (aegis_engine_core_v2/financial_filters.py, financial_analysis.py, harmonic_analysis.py -- NumPy/SciPy
only) on seeded inputs.  Build container only; /root/reference does not travel.

The package __init__ of aegis_engine_core_v2 imports librosa (absent), so the modules are loaded one by one
under a synthetic parent package.  `detect_slides_macd` calls librosa.hz_to_midi: a stub module supplying only that
one formula (12*(log2(f) - log2(440)) + 69) is registered, so the "slides" goldens depend on that restatement.
harmonic_analysis imports librosa but never calls it."""
import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference/aegis_engine_core_v2"
HERE = os.path.dirname(os.path.abspath(__file__))
ART = {None: 0, "normal": 1, "bend": 2, "vibrato": 3, "noise": 4}
SLD = {None: 0, "normal": 1, "slide_up": 2, "slide_down": 3}


def load_reference():
    lib = types.ModuleType("librosa")
    lib.hz_to_midi = lambda f: 12 * (np.log2(np.asanyarray(f)) - np.log2(440.0)) + 69
    sys.modules["librosa"] = lib
    pkg = types.ModuleType("aegis_engine_core_v2")
    pkg.__path__ = [REF]
    sys.modules["aegis_engine_core_v2"] = pkg
    mods = {}
    for name in ("financial_filters", "financial_analysis", "harmonic_analysis"):
        spec = importlib.util.spec_from_file_location(f"aegis_engine_core_v2.{name}", os.path.join(REF, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods


def make_tracks():
    rng = np.random.default_rng(20260220)
    tracks = {}

    def melody(n, nan_frac=0.0, lead=0, vib=0.0):
        f = np.empty(n)
        i = 0
        while i < n:
            L = int(rng.integers(3, 40))
            midi = rng.integers(40, 85)
            hz = 440.0 * 2 ** ((midi - 69) / 12)
            seg = hz * (1 + vib * np.sin(np.arange(L) * 0.9)) + rng.normal(0, 0.3, L)
            f[i:i + L] = seg[: n - i]
            i += L
        if nan_frac:
            j = 0
            while j < n:
                j += int(rng.integers(2, 30))
                g = int(rng.integers(1, max(2, int(20 * nan_frac * 4))))
                f[j:j + g] = np.nan
                j += g
        f[:lead] = np.nan
        return f

    tracks["dense_200"] = melody(200)
    tracks["gaps_300"] = melody(300, nan_frac=0.25)
    tracks["lead_nan_120"] = melody(120, nan_frac=0.1, lead=17)
    tracks["vibrato_400"] = melody(400, nan_frac=0.05, vib=0.02)
    tracks["long_2000"] = melody(2000, nan_frac=0.3, vib=0.01)
    tracks["short_30"] = melody(30, nan_frac=0.2)
    tracks["tiny_12"] = melody(12)
    tracks["exact_26"] = melody(26, nan_frac=0.1)
    tracks["all_nan_40"] = np.full(40, np.nan)
    one = np.full(25, np.nan); one[7] = 220.0
    tracks["one_valid_25"] = one
    two = np.full(25, np.nan); two[3] = 220.0; two[19] = 233.1
    tracks["two_valid_25"] = two
    tracks["constant_64"] = np.full(64, 196.0)
    dens = np.concatenate([rng.integers(0, 4, 60).astype(float), np.zeros(10), rng.integers(0, 7, 50).astype(float)])
    tracks["density_120"] = dens
    tracks["density_10"] = rng.integers(0, 3, 10).astype(float)
    return tracks


def main():
    mods = load_reference()
    FA, FF, HA = mods["financial_analysis"], mods["financial_filters"], mods["harmonic_analysis"]
    an = FA.FinancialPitchAnalyzer(sr=22050, hop_length=512)
    out = {}
    names = []
    for name, x in make_tracks().items():
        names.append(name)
        out[f"{name}/x"] = x
        for w in (5, 10, 20):
            if len(x) >= w:
                out[f"{name}/sma{w}"] = an.simple_moving_average(x, window=w)
        for span in (5, 12, 26):
            out[f"{name}/ema{span}"] = an.exponential_moving_average(x, span=span)
        for w, k in ((10, 2.0), (20, 2)):
            if len(x) >= w:
                ma, up, lo = an.bollinger_bands(x, window=w, num_std=k)
                out[f"{name}/boll{w}_ma"], out[f"{name}/boll{w}_up"], out[f"{name}/boll{w}_lo"] = ma, up, lo
        if len(x) >= 10:
            out[f"{name}/artic"] = np.array([ART[a] for a in an.detect_articulation_bollinger(x, window=10, sensitivity=2.0)], np.int8)
        m, s, h = an.macd(x, fast=12, slow=26, signal=9)
        out[f"{name}/macd"], out[f"{name}/macd_sig"], out[f"{name}/macd_hist"] = m, s, h
        if not name.startswith("density"):
            for thr in (0.5, 0.3):
                out[f"{name}/slides{thr}"] = np.array([SLD[a] for a in an.detect_slides_macd(x, threshold=thr)], np.int8)
        xr = np.nan_to_num(x) if name.startswith("density") else np.nan_to_num(x) / 100.0
        for per in (14, 5):
            out[f"{name}/rsi{per}"] = an.rsi(xr, period=per)
        out[f"{name}/savgol"] = np.array(FF.FinancialNoiseFilters.savitzky_golay(x.copy()))
        out[f"{name}/kalman"] = np.array(FF.FinancialNoiseFilters.kalman_filter(x.copy()))
        out[f"{name}/holt"] = np.array(FF.FinancialNoiseFilters.holt_winters(x.copy()))
        med, conf = FF.multi_filter_consensus(x.copy())
        out[f"{name}/cons_med"], out[f"{name}/cons_conf"] = med, conf
        if len(x) >= 10 and not name.startswith("density"):
            for adv in (True, False):
                r = an.analyze_pitch_financial(x.copy(), ~np.isnan(x), use_advanced_filters=adv)
                tag = "adv" if adv else "ema"
                out[f"{name}/apf_{tag}_trend"] = np.array(r["trend"])
                out[f"{name}/apf_{tag}_artic"] = np.array([ART[a] for a in r["articulations"]], np.int8)
                out[f"{name}/apf_{tag}_slides"] = np.array([SLD[a] for a in r["slides"]], np.int8)
                out[f"{name}/apf_{tag}_conf"] = np.array(r["confidence"])
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "v2_trend_golden.npz"), **out)

    # ghost-note RSI filter + harmonic analysis on note lists
    rng = np.random.default_rng(7)
    hz = HA.HarmonicAnalyzer()
    cases = []
    note_sets = {
        "c_major": [60, 62, 64, 65, 67, 69, 71, 72], "a_minor": [57, 59, 60, 62, 64, 65, 67, 69],
        "c_blues": [60, 63, 65, 66, 67, 70, 72], "noisy": [60, 61, 62, 63, 64, 65, 66, 67, 68, 69], "empty": [],
        "e_minor_riff": [40, 43, 45, 47, 50, 52, 55, 52, 50, 47, 45, 43, 40, 41],
    }
    for i in range(6):
        note_sets[f"rand{i}"] = [int(v) for v in rng.integers(36, 90, int(rng.integers(1, 60)))]
    for name, notes in note_sets.items():
        midi = np.array(notes, dtype=int)
        conf = rng.uniform(0.2, 1.0, len(midi)).round(6)
        times = np.sort(rng.uniform(0, 9000, len(midi))).round(3)
        durs = rng.uniform(50, 800, len(midi)).round(3)
        rec = {"name": name, "midi": notes, "conf": conf.tolist(), "times": times.tolist(), "durs": durs.tolist()}
        rec["key"] = hz.detect_key(midi)
        rec["key_dur"] = hz.detect_key(midi, use_duration=True, durations=durs)
        if len(midi):
            for tol in (0, 1, 2):
                fm, fc, mask = hz.filter_out_of_scale_notes(midi, conf, rec["key"], tolerance=tol)
                rec[f"mask{tol}"] = mask.astype(int).tolist()
            rec["chords"] = hz.analyze_chord_progression(midi, times)
            rec["adaptive"] = np.asarray(hz.adaptive_filter_by_context(midi, times, conf, rec["key"])).tolist()
            import contextlib, io
            with contextlib.redirect_stdout(io.StringIO()):
                r = HA.apply_harmonic_filter(midi, conf, times=times, tolerance=1)
            rec["apply"] = {"filtered_midi": np.asarray(r["filtered_midi"]).tolist(),
                            "filtered_confidence": np.asarray(r["filtered_confidence"]).tolist(),
                            "out_of_scale_mask": np.asarray(r["out_of_scale_mask"]).astype(int).tolist()}
            events = [{"start": float(t) / 1000, "end": float(t + d) / 1000, "note": int(n)} for n, t, d in zip(midi, times, durs)]
            kept = an.filter_ghost_notes_rsi(events, rsi_threshold=70)
            rec["ghost_kept"] = [events.index(e) for e in kept]
        cases.append(rec)
    json.dump(cases, open(os.path.join(HERE, "v2_harmonic_golden.json"), "w"), indent=0, default=float)
    print("wrote", len(names), "trend tracks,", len(cases), "note sets")


if __name__ == "__main__":
    main()
