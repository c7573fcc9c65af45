"""v2 "financial" pitch analysis on the GPU: the reference's `FinancialPitchAnalyzer`
(/root/reference/aegis_engine_core_v2/financial_analysis.py:25-423) and `FinancialNoiseFilters` /
`multi_filter_consensus` (financial_filters.py:25-141, 256-298) with the same class, method and keyword
names, return shapes and quirks; every array operation is one `aegis_trend` call (csrc/trend.hip)."""
import numpy as np
import scipy.signal

from . import _lib

_ARTICULATIONS = (None, "normal", "bend", "vibrato", "noise")
_SLIDES = (None, "normal", "slide_up", "slide_down")
_handles = {}


def _handle(device=0):
    if device not in _handles:
        _handles[device] = _lib.Handle(device=device, scipy_tables=False)
    return _handles[device]


def _one(op, data, params, n_out=1, dtype=np.float64, device=0):
    res = _handle(device).trend(op, [np.asarray(data, dtype=np.float64)], params, n_out=n_out, out_dtype=dtype)
    return [r[0] for r in res]


class FinancialNoiseFilters:
    """financial_filters.py:17-141 (the dead-code ATR / Ichimoku / stochastic helpers are not part of the
    analyze path and are not provided)."""
    device = 0

    @staticmethod
    def savitzky_golay(data, window=11, polyorder=3):
        data = np.asarray(data, dtype=np.float64)
        # scipy.signal.savgol_filter = correlate with the reversed coefficients; ndimage folds the
        # two sides when the kernel is symmetric to DBL_EPSILON (ni_filters.c) -- same test here
        coef = scipy.signal.savgol_coeffs(window, polyorder)[::-1]
        symmetric = bool(np.all(np.abs(coef - coef[::-1]) <= np.finfo(float).eps))
        return _one(_lib.TREND_SAVGOL, data, [window, int(symmetric), *coef], device=FinancialNoiseFilters.device)[0]

    @staticmethod
    def kalman_filter(data, process_variance=1e-5, measurement_variance=1e-1):
        return _one(_lib.TREND_KALMAN, data, [process_variance, measurement_variance], device=FinancialNoiseFilters.device)[0]

    @staticmethod
    def holt_winters(data, alpha=0.3, beta=0.1):
        return _one(_lib.TREND_HOLT, data, [alpha, beta], device=FinancialNoiseFilters.device)[0]


def multi_filter_consensus(data, filters=("savgol", "kalman", "holt")):
    """-> (nanmedian of the filter outputs, 1 / (1 + nanstd))  (financial_filters.py:256-298)."""
    data = np.asarray(data, dtype=np.float64)
    f = FinancialNoiseFilters
    results = []
    if "savgol" in filters:
        results.append(f.savitzky_golay(data))
    if "kalman" in filters:
        results.append(f.kalman_filter(data))
    if "holt" in filters:
        results.append(f.holt_winters(data))
    if not results:
        return data, np.ones_like(data)
    if len(data) == 0:
        return data.copy(), np.ones_like(data)
    med, conf = _handle(f.device).trend(_lib.TREND_CONSENSUS, np.array(results), [len(results)], n_out=2,
                                        stacked_rows=len(results))
    return med[0], conf[0]


class FinancialPitchAnalyzer:
    def __init__(self, sr=22050, hop_length=512, device=0):
        self.sr = sr
        self.hop_length = hop_length
        self.ms_per_frame = (hop_length / sr) * 1000
        self.device = device

    def _op(self, op, data, params, n_out=1, dtype=np.float64):
        return _one(op, data, params, n_out, dtype, self.device)

    def simple_moving_average(self, data, window=5):
        data = np.asarray(data, dtype=np.float64)
        if len(data) < window:      # np.convolve(..., 'same') returns `window` samples and the NaN restore fails
            raise IndexError(f"boolean index did not match indexed array along axis 0; size of axis is {window} "
                             f"but size of corresponding boolean axis is {len(data)}")
        return self._op(_lib.TREND_SMA, data, [window])[0]

    def exponential_moving_average(self, data, span=5):
        return self._op(_lib.TREND_EMA, data, [span])[0]

    def bollinger_bands(self, data, window=20, num_std=2):
        data = np.asarray(data, dtype=np.float64)
        if len(data) < window:
            raise IndexError(f"series of {len(data)} samples is shorter than the window {window}")
        return tuple(self._op(_lib.TREND_BOLLINGER, data, [window, num_std], n_out=3))

    def detect_articulation_bollinger(self, f0, window=10, sensitivity=2.0):
        f0 = np.asarray(f0, dtype=np.float64)
        if len(f0) < window:
            raise IndexError(f"series of {len(f0)} samples is shorter than the window {window}")
        codes = self._op(_lib.TREND_ARTICULATION, f0, [window, sensitivity], dtype=np.int8)[0]
        return [_ARTICULATIONS[c] for c in codes]

    def macd(self, data, fast=12, slow=26, signal=9):
        return tuple(self._op(_lib.TREND_MACD, data, [fast, slow, signal], n_out=3))

    def detect_slides_macd(self, f0, threshold=0.5):
        codes = self._op(_lib.TREND_SLIDES, f0, [threshold], dtype=np.int8)[0]
        return [_SLIDES[c] for c in codes]

    def rsi(self, data, period=14):
        return self._op(_lib.TREND_RSI, data, [period])[0]

    @staticmethod
    def _ghost_density(note_events):
        """The note-density track of filter_ghost_notes_rsi (financial_analysis.py:333-347): +1 over [start*10, end*10) per
        note.  Counts are small integers, so the difference-array form below gives the same float64 values as the
        reference's slice increments."""
        starts = np.fromiter((e["start"] for e in note_events), dtype=np.float64, count=len(note_events))
        ends = np.fromiter((e["end"] for e in note_events), dtype=np.float64, count=len(note_events))
        max_time = ends.max()
        n = int(max_time * 10)                               # len(np.linspace(0, max_time, int(max_time * 10)))
        a, b = (starts * 10).astype(np.int64), (ends * 10).astype(np.int64)
        ok = a < n
        b = np.minimum(b, n)
        ok &= b > a
        diff = np.zeros(n + 1)
        np.add.at(diff, a[ok], 1.0)
        np.add.at(diff, b[ok], -1.0)
        return np.cumsum(diff[:-1]) if n > 0 else np.zeros(0)

    @staticmethod
    def _ghost_keep(note_events, rsi_values, rsi_threshold):
        kept = []
        for e in note_events:
            i = int(e["start"] * 10)
            if i >= len(rsi_values) or rsi_values[i] < rsi_threshold:
                kept.append(e)
        return kept

    def filter_ghost_notes_rsi(self, note_events, rsi_threshold=70):
        """financial_analysis.py:322-362, including its mixed units: `start`/`end` are scaled by 10 as if
        they were seconds whatever the caller stores there."""
        if not note_events:
            return note_events
        density = self._ghost_density(note_events)
        rsi_values = self.rsi(density, period=14) if len(density) else np.zeros(0)
        return self._ghost_keep(note_events, rsi_values, rsi_threshold)

    def filter_ghost_notes_rsi_batch(self, event_lists, rsi_threshold=70):
        """filter_ghost_notes_rsi for several clips: every clip's density track is one series of ONE library call, which
        returns the two Wilder averages; the RSI value (avg_gain / avg_loss -> 100 - 100 / (1 + rs), the reference's
        operations) is then formed at the few positions the filter reads."""
        out = list(event_lists)
        live = [i for i, ev in enumerate(event_lists) if ev]
        if not live:
            return out
        # int(start * 10), int(end * 10) of every note and int(max_end * 10) per clip: all the library needs to build the
        # density tracks on the device (aegis_ghost_rsi); the tracks themselves (77 k elements per three-minute clip)
        # never exist on the host
        counts = [len(event_lists[i]) for i in live]
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        starts = np.fromiter((e["start"] for i in live for e in event_lists[i]), dtype=np.float64, count=int(off[-1]))
        ends = np.fromiter((e["end"] for i in live for e in event_lists[i]), dtype=np.float64, count=int(off[-1]))
        a, b = (starts * 10).astype(np.int64), (ends * 10).astype(np.int64)
        n = np.array([int(ends[off[j]:off[j + 1]].max() * 10) for j in range(len(live))], dtype=np.int64)
        g_all, l_all = _handle(self.device).ghost_rsi(a, b, off, n, period=14)
        for j, i in enumerate(live):
            ev = event_lists[i]
            sl = slice(int(off[j]), int(off[j + 1]))
            inside = a[sl] < n[j]
            g, l = g_all[sl][inside], l_all[sl][inside]
            with np.errstate(divide="ignore", invalid="ignore"):
                val = 100 - (100 / (1 + g / l))
            val = np.where(l == 0, 100.0, val)
            val = np.where(np.isnan(g), 50.0, val)            # the first `period` positions and tracks shorter than it
            keep = np.ones(len(ev), bool)
            keep[inside] = val < rsi_threshold
            out[i] = [e for e, k in zip(ev, keep) if k]
        return out

    def analyze_pitch_financial(self, f0, voiced_flag, use_advanced_filters=True):
        """-> {'trend', 'articulations', 'slides', 'confidence'}  (financial_analysis.py:368-423)."""
        f0 = np.asarray(f0, dtype=np.float64)
        if use_advanced_filters:
            return self.analyze_pitch_financial_batch([f0])[0]
        trend = self.exponential_moving_average(f0, span=5)
        articulations = self.detect_articulation_bollinger(f0, window=10)
        slides = self.detect_slides_macd(f0, threshold=0.3)
        _, upper, lower = self.bollinger_bands(f0, window=10)
        width = upper - lower
        ok = ~np.isnan(f0) & ~np.isnan(width)
        confidence = np.zeros_like(f0)
        confidence[ok] = np.where(width[ok] > 0, 1.0 / (1.0 + width[ok]), 1.0)
        return {"trend": trend, "articulations": articulations, "slides": slides, "confidence": confidence}

    def analyze_pitch_financial_batch(self, tracks, labels=True):
        """analyze_pitch_financial (use_advanced_filters=True) for a list of pitch tracks in ONE library call
        (AEGIS_TREND_PITCH_ANALYSIS: the consensus filters, both state machines and the band-width confidence, the four
        sequential walks on four streams at once).  labels=False leaves 'articulations' / 'slides' as int8 code arrays
        (indices into _ARTICULATIONS / _SLIDES) instead of the reference's lists of strings."""
        tracks = [np.asarray(t, dtype=np.float64) for t in tracks]
        for t in tracks:
            if len(t) < 10:        # bollinger_bands(window=10) on a shorter track raises in the reference (np.convolve 'same')
                raise IndexError(f"series of {len(t)} samples is shorter than the window 10")
        coef = scipy.signal.savgol_coeffs(11, 3)[::-1]
        symmetric = bool(np.all(np.abs(coef - coef[::-1]) <= np.finfo(float).eps))
        params = [11, int(symmetric), *coef, 1e-5, 1e-1, 0.3, 0.1, 10, 2, 0.3]
        trend, art, sl, conf = _handle(self.device).trend(_lib.TREND_PITCH_ANALYSIS, tracks, params, n_out=4,
                                                           out_dtype=[np.float64, np.int8, np.int8, np.float64])
        out = []
        for i in range(len(tracks)):
            a, s_ = art[i], sl[i]
            if labels:
                a, s_ = [_ARTICULATIONS[c] for c in a], [_SLIDES[c] for c in s_]
            out.append({"trend": trend[i], "articulations": a, "slides": s_, "confidence": conf[i]})
        return out
