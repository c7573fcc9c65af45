"""v2 trend filters on the GPU (aegis_trend) against goldens produced by the
reference's own financial_analysis.py / financial_filters.py -- PINNED.  Sequential recurrences (EMA, MACD,
RSI, Kalman, Holt) and the state machines are expected bit-exact; windowed sums within 1e-12 relative (the
reference's np.convolve / BLAS summation order is not specified)."""
import json
import os

import numpy as np
import pytest

from spectrogram_midi_amd import _lib
from spectrogram_midi_amd.financial import FinancialNoiseFilters, FinancialPitchAnalyzer, multi_filter_consensus

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "v2_trend_golden.npz"))
NAMES = [str(n) for n in G["names"]]
ART = (None, "normal", "bend", "vibrato", "noise")
SLD = (None, "normal", "slide_up", "slide_down")
an = FinancialPitchAnalyzer(sr=22050, hop_length=512)


def close(a, b, tag, rtol=1e-12, atol=1e-12):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, equal_nan=True, err_msg=tag)


def exact(a, b, tag):
    np.testing.assert_array_equal(a, b, err_msg=tag)      # NaNs compare equal position-wise


def test_moving_averages():
    for n in NAMES:
        x = G[f"{n}/x"]
        for w in (5, 10, 20):
            if f"{n}/sma{w}" in G:
                close(an.simple_moving_average(x, window=w), G[f"{n}/sma{w}"], f"{n} sma{w}")
            else:
                with pytest.raises(IndexError):
                    an.simple_moving_average(x, window=w)
        for span in (5, 12, 26):
            exact(an.exponential_moving_average(x, span=span), G[f"{n}/ema{span}"], f"{n} ema{span}")


def test_bollinger_and_articulations():
    for n in NAMES:
        x = G[f"{n}/x"]
        for w, k in ((10, 2.0), (20, 2)):
            if f"{n}/boll{w}_ma" not in G:
                continue
            ma, up, lo = an.bollinger_bands(x, window=w, num_std=k)
            close(ma, G[f"{n}/boll{w}_ma"], f"{n} ma{w}")
            close(up, G[f"{n}/boll{w}_up"], f"{n} up{w}")
            close(lo, G[f"{n}/boll{w}_lo"], f"{n} lo{w}")
        if f"{n}/artic" in G:
            got = an.detect_articulation_bollinger(x, window=10, sensitivity=2.0)
            assert got == [ART[c] for c in G[f"{n}/artic"]], n


def test_macd_slides_rsi():
    for n in NAMES:
        x = G[f"{n}/x"]
        m, s, h = an.macd(x, fast=12, slow=26, signal=9)
        exact(m, G[f"{n}/macd"], f"{n} macd"); exact(s, G[f"{n}/macd_sig"], f"{n} sig"); exact(h, G[f"{n}/macd_hist"], f"{n} hist")
        for thr in (0.5, 0.3):
            if f"{n}/slides{thr}" in G:
                assert an.detect_slides_macd(x, threshold=thr) == [SLD[c] for c in G[f"{n}/slides{thr}"]], (n, thr)
        xr = np.nan_to_num(x) if n.startswith("density") else np.nan_to_num(x) / 100.0
        for per in (14, 5):
            exact(an.rsi(xr, period=per), G[f"{n}/rsi{per}"], f"{n} rsi{per}")


def test_noise_filters_and_consensus():
    for n in NAMES:
        x = G[f"{n}/x"]
        close(FinancialNoiseFilters.savitzky_golay(x), G[f"{n}/savgol"], f"{n} savgol")
        exact(FinancialNoiseFilters.kalman_filter(x), G[f"{n}/kalman"], f"{n} kalman")
        exact(FinancialNoiseFilters.holt_winters(x), G[f"{n}/holt"], f"{n} holt")
        med, conf = multi_filter_consensus(x)
        close(med, G[f"{n}/cons_med"], f"{n} median")
        close(conf, G[f"{n}/cons_conf"], f"{n} conf", rtol=1e-9)


def test_analyze_pitch_financial():
    for n in NAMES:
        if f"{n}/apf_adv_trend" not in G:
            continue
        x = G[f"{n}/x"]
        for adv, tag in ((True, "adv"), (False, "ema")):
            r = an.analyze_pitch_financial(x, ~np.isnan(x), use_advanced_filters=adv)
            close(r["trend"], G[f"{n}/apf_{tag}_trend"], f"{n} trend {tag}")
            assert r["articulations"] == [ART[c] for c in G[f"{n}/apf_{tag}_artic"]]
            assert r["slides"] == [SLD[c] for c in G[f"{n}/apf_{tag}_slides"]]
            close(r["confidence"], G[f"{n}/apf_{tag}_conf"], f"{n} conf {tag}", rtol=1e-9)


def test_ghost_rsi_on_the_device_equals_the_rsi_of_the_density_tracks():
    """aegis_ghost_rsi builds the density tracks from the notes' intervals on the device and returns the Wilder averages at
    the notes' own positions: bit-identical to AEGIS_TREND_RSI (averages) on the tracks the host would have built
    (FinancialPitchAnalyzer._ghost_density), including notes that start at the end of the track, tracks shorter than the
    period, a clip whose notes all end before 0.1 and frame-unit times (tracks of tens of thousands of elements)."""
    from spectrogram_midi_amd.financial import _handle
    rng = np.random.default_rng(11)
    lists = []
    for n_notes, span in ((40, 300.0), (12, 7752.0), (300, 7752.0), (3, 1.0), (2, 0.05), (25, 60.0)):
        st = np.sort(rng.uniform(0, span, n_notes))
        lists.append([{"start": float(a), "end": float(a + rng.uniform(0.001, max(span / 10, 0.002)))} for a in st])
    lists[1][-1]["start"] = lists[1][-1]["end"] = max(e["end"] for e in lists[1])          # a note at the very end of its track
    h = _handle(0)
    off = np.concatenate([[0], np.cumsum([len(ev) for ev in lists])]).astype(np.int64)
    starts = np.array([e["start"] for ev in lists for e in ev]); ends = np.array([e["end"] for ev in lists for e in ev])
    a, b = (starts * 10).astype(np.int64), (ends * 10).astype(np.int64)
    n = np.array([int(max(e["end"] for e in ev) * 10) for ev in lists], dtype=np.int64)
    g, l = h.ghost_rsi(a, b, off, n)
    for j, ev in enumerate(lists):
        dens = an._ghost_density(ev)
        assert len(dens) == n[j]
        sl = slice(int(off[j]), int(off[j + 1]))
        inside = a[sl] < n[j]
        assert np.isnan(g[sl][~inside]).all() and np.isnan(l[sl][~inside]).all()
        if n[j] == 0:
            continue
        ag, al = h.trend(_lib.TREND_RSI, [dens], [14, 1], n_out=2)
        np.testing.assert_array_equal(g[sl][inside], ag[0][a[sl][inside]])
        np.testing.assert_array_equal(l[sl][inside], al[0][a[sl][inside]])


def test_ghost_note_filter_matches_reference():
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "v2_harmonic_golden.json")))
    lists, want = [], []
    for c in cases:
        if not c["midi"]:
            continue
        events = [{"start": float(t) / 1000, "end": float(t + d) / 1000, "note": int(n)}
                  for n, t, d in zip(c["midi"], c["times"], c["durs"])]
        kept = an.filter_ghost_notes_rsi(events, rsi_threshold=70)
        assert [events.index(e) for e in kept] == c["ghost_kept"], c["name"]
        lists.append(events)
        want.append(c["ghost_kept"])
    # all the clips' density tracks in ONE call, the RSI formed from the two Wilder averages at the positions read
    for events, kept, w in zip(lists, an.filter_ghost_notes_rsi_batch(lists + [[]])[:-1], want):
        assert [events.index(e) for e in kept] == w
    assert an.filter_ghost_notes_rsi([]) == []


def test_ragged_batch_equals_single_series():
    h = _lib.Handle(scipy_tables=False)
    series = [G[f"{n}/x"] for n in NAMES if len(G[f"{n}/x"]) >= 26]
    batch = h.trend(_lib.TREND_EMA, series, [12])[0]
    for s, b in zip(series, batch):
        exact(b, an.exponential_moving_average(s, span=12), "batch ema")
    k = h.trend(_lib.TREND_KALMAN, series, [1e-5, 1e-1])[0]
    for s, b in zip(series, k):
        exact(b, FinancialNoiseFilters.kalman_filter(s), "batch kalman")
    with pytest.raises(_lib.AegisError):
        h.trend(99, series, [1])
    h.close()


def test_fused_pitch_analysis_batch_equals_the_single_ops():
    """AEGIS_TREND_PITCH_ANALYSIS (analyze_pitch_financial as one call, four streams) over a ragged batch == the single
    ops series by series, bit for bit: trend = consensus median of savgol / kalman / holt, Bollinger articulation codes,
    MACD slide codes, band-width confidence."""
    series = [G[f"{n}/x"] for n in NAMES if len(G[f"{n}/x"]) >= 26]
    assert len(series) >= 3
    res = an.analyze_pitch_financial_batch(series)
    raw = an.analyze_pitch_financial_batch(series, labels=False)
    for s, r, q in zip(series, res, raw):
        med, _ = multi_filter_consensus(s)
        exact(r["trend"], med, "fused trend")
        assert r["articulations"] == an.detect_articulation_bollinger(s, window=10)
        assert r["slides"] == an.detect_slides_macd(s, threshold=0.3)
        _, up, lo = an.bollinger_bands(s, window=10)
        w = up - lo
        ok = ~np.isnan(s) & ~np.isnan(w)
        conf = np.zeros_like(s)
        conf[ok] = np.where(w[ok] > 0, 1.0 / (1.0 + w[ok]), 1.0)
        exact(r["confidence"], conf, "fused confidence")
        assert q["articulations"].dtype == np.int8 and [ART[c] for c in q["articulations"]] == r["articulations"]
    with pytest.raises(IndexError):
        an.analyze_pitch_financial_batch([np.ones(5)])
