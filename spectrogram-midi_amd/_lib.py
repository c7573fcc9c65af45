"""ctypes binding of libaegis_hip.so (include/aegis_hip.h).  Fails loudly when the
extension has not been built: `python -c "import __graft_entry__ as g; g.build()"`."""
import atexit
import ctypes as C
import os
import sys
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AEGIS_HIP_LIB", os.path.join(_HERE, "libaegis_hip.so"))

STAGE_MEL, STAGE_RAKE, STAGE_PYIN, STAGE_RMS, STAGE_ALL = 0x1, 0x2, 0x4, 0x8, 0xF
OPT_CHECK_FINITE, OPT_F0_ZERO = 0x10, 0x20
(TREND_SMA, TREND_EMA, TREND_BOLLINGER, TREND_ARTICULATION, TREND_MACD, TREND_SLIDES, TREND_RSI, TREND_SAVGOL,
 TREND_KALMAN, TREND_HOLT, TREND_CONSENSUS, TREND_PITCH_ANALYSIS) = range(1, 13)
OK, ERR_INVALID, ERR_NOMEM, ERR_DEVICE, ERR_UNSUPPORTED = 0, -22, -12, -5, -95
PYIN_INIT_UNVOICED, PYIN_INIT_UNIFORM = 0, 1      # aegis_config.pyin_init
ABI_VERSION = 2                                   # include/aegis_hip.h AEGIS_ABI_VERSION: the layout Config / Outputs below assume
_PYIN_INIT = {"unvoiced": 0, "uniform": 1, 0: 0, 1: 1}


class AegisError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libaegis_hip error {code}: {msg}")
        self.code = code


class Config(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("hop_length", C.c_int32), ("n_fft", C.c_int32),
                ("n_mels", C.c_int32), ("fmin", C.c_double), ("fmax", C.c_double),
                ("device", C.c_int32), ("pyin_init", C.c_int32), ("max_frames_per_pass", C.c_int64)]


class StreamFrames(C.Structure):
    _fields_ = [("rms", C.c_void_p), ("voiced_prob", C.c_void_p), ("live_state", C.c_void_p)]


class Outputs(C.Structure):
    _fields_ = [("f0", C.c_void_p), ("voiced_flag", C.c_void_p), ("voiced_prob", C.c_void_p),
                ("rms", C.c_void_p), ("rake_mask", C.c_void_p), ("S_dB", C.c_void_p), ("pitch_bin", C.c_void_p), ("sdb_col_means", C.c_void_p)]


EXPORTS = ("aegis_abi_version", "aegis_create", "aegis_destroy", "aegis_last_error", "aegis_frames_for",
           "aegis_analyze_batch", "aegis_analyze_batch_device", "aegis_get_table", "aegis_get_param",
           "aegis_debug_fetch", "aegis_set_profiling", "aegis_last_kernel_ms", "aegis_rake_patterns", "aegis_set_table", "aegis_last_kernel_launches", "aegis_trend", "aegis_ghost_rsi",
           "aegis_stream_open", "aegis_stream_push", "aegis_stream_close", "aegis_stream_free", "aegis_cqt", "aegis_cqt_device", "aegis_chroma_cqt",
           "aegis_extract_events", "aegis_render_smf", "aegis_events_last_error")

_lib = None


def load():
    """Returns the loaded library; raises ImportError with build instructions if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension must be built (make -C {_HERE}/csrc, or "
            "__graft_entry__.build()).  There is no CPU fallback for the analyze path.")
    lib = C.CDLL(LIB_PATH)
    lib.aegis_abi_version.restype = C.c_int
    got = int(lib.aegis_abi_version())
    if got != ABI_VERSION:
        # aegis_outputs / aegis_config carry no size field: a library of another ABI would read this module's structs at
        # the wrong offsets and write device or host memory through stale pointers (AEGIS_HIP_LIB is routinely pointed at
        # other builds)
        raise ImportError(f"{LIB_PATH} has ABI version {got}, this binding was written for {ABI_VERSION} "
                          "(include/aegis_hip.h AEGIS_ABI_VERSION): rebuild the library or update the binding")
    lib.aegis_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    lib.aegis_create.restype = C.c_int
    lib.aegis_destroy.argtypes = [C.c_void_p]
    lib.aegis_destroy.restype = None
    lib.aegis_last_error.argtypes = [C.c_void_p]
    lib.aegis_last_error.restype = C.c_char_p
    lib.aegis_frames_for.argtypes = [C.c_void_p, C.c_int64]
    lib.aegis_frames_for.restype = C.c_int64
    lib.aegis_analyze_batch.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int32,
                                        C.c_double, C.c_uint32, C.POINTER(Outputs)]
    lib.aegis_analyze_batch.restype = C.c_int
    lib.aegis_analyze_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32,
                                               C.c_double, C.c_uint32, C.POINTER(Outputs), C.c_void_p, C.c_int32]
    lib.aegis_analyze_batch_device.restype = C.c_int
    lib.aegis_rake_patterns.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_void_p]
    lib.aegis_rake_patterns.restype = C.c_int
    lib.aegis_trend.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                C.POINTER(C.c_void_p), C.c_int32]
    lib.aegis_trend.restype = C.c_int
    lib.aegis_ghost_rsi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.aegis_ghost_rsi.restype = C.c_int
    lib.aegis_stream_open.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
    lib.aegis_stream_open.restype = C.c_int
    lib.aegis_stream_push.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(StreamFrames), C.POINTER(C.c_int64)]
    lib.aegis_stream_push.restype = C.c_int
    lib.aegis_stream_close.argtypes = [C.c_void_p, C.c_double, C.POINTER(Outputs), C.POINTER(C.c_int64)]
    lib.aegis_stream_close.restype = C.c_int
    lib.aegis_stream_free.argtypes = [C.c_void_p]
    lib.aegis_stream_free.restype = None
    lib.aegis_cqt.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.c_int32,
                              C.c_double, C.c_double, C.c_void_p]
    lib.aegis_cqt.restype = C.c_int
    lib.aegis_cqt_device.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.c_int32, C.c_double,
                                     C.c_double, C.c_void_p, C.c_void_p, C.c_int32]
    lib.aegis_cqt_device.restype = C.c_int
    lib.aegis_chroma_cqt.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.c_int32,
                                     C.c_double, C.c_double, C.c_int32, C.c_void_p, C.c_void_p]
    lib.aegis_chroma_cqt.restype = C.c_int
    lib.aegis_set_table.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]
    lib.aegis_set_table.restype = C.c_int
    lib.aegis_get_table.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]
    lib.aegis_get_table.restype = C.c_int64
    lib.aegis_get_param.argtypes = [C.c_void_p, C.c_char_p]
    lib.aegis_get_param.restype = C.c_int64
    lib.aegis_debug_fetch.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]
    lib.aegis_debug_fetch.restype = C.c_int64
    lib.aegis_set_profiling.argtypes = [C.c_void_p, C.c_int32]
    lib.aegis_set_profiling.restype = C.c_int
    lib.aegis_last_kernel_ms.argtypes = [C.c_void_p, C.c_char_p]
    lib.aegis_last_kernel_ms.restype = C.c_double
    lib.aegis_last_kernel_launches.argtypes = [C.c_void_p, C.c_char_p]
    lib.aegis_last_kernel_launches.restype = C.c_int
    _lib = lib
    return lib


_TABLE_DTYPES = {"mel_dense": np.float32}
_DEBUG_DTYPES = {"persistent_fallbacks": np.int64, "obs_cycles": np.int64, "cqt_cycles": np.int64, "viterbi_cycles": np.int64, "viterbi_spans": np.int64, "split_verify": np.int64, "split_flags": np.int64, "seg_lock": np.int64, "frame_cycles": np.int64, "states": np.int32, "melpow": np.float32}


_live_handles = weakref.WeakSet()


@atexit.register
def _close_live_handles():
    """Handles still alive at interpreter exit (a module global, a handle held by the traceback of the exception that is
    ending the process) are closed here, while the interpreter and the HIP runtime are both fully alive: atexit callbacks
    run before module teardown.  `Handle.__del__` does nothing once finalisation has begun."""
    for h in list(_live_handles):
        try:
            h.close()
        except Exception:       # noqa: BLE001 -- exit must go on
            pass


class Handle:
    """One analyze context (tables + workspace + stream) on one GPU.  device=-1 builds the
    host tables only (no GPU touched).  pyin_init: "unvoiced" (default; librosa core/pitch.py::pyin: p_init zero on
    the voiced half, 1/B on the unvoiced half) or "uniform" (1/(2B) everywhere, SURVEY.md P11's reading)."""

    def __init__(self, sample_rate=44100, hop_length=512, n_fft=2048, n_mels=128, fmin=0.0, fmax=0.0,
                 device=0, max_frames_per_pass=0, scipy_tables=True, pyin_init="unvoiced"):
        self.lib = load()
        if pyin_init not in _PYIN_INIT:
            raise ValueError("pyin_init must be 'unvoiced' (librosa's p_init) or 'uniform'")
        self.pyin_init = "uniform" if _PYIN_INIT[pyin_init] else "unvoiced"
        cfg = Config(sample_rate, hop_length, n_fft, n_mels, fmin, fmax, device, _PYIN_INIT[pyin_init], max_frames_per_pass)
        h = C.c_void_p()
        rc = self.lib.aegis_create(C.byref(cfg), C.byref(h))
        if rc != OK:
            raise AegisError(rc, self.lib.aegis_last_error(None).decode())
        self._h = h
        _live_handles.add(self)
        self.sr, self.hop, self.n_fft, self.n_mels, self.device = sample_rate, hop_length, n_fft, n_mels, device
        if scipy_tables:
            self._load_scipy_tables()

    def _load_scipy_tables(self):
        """The pYIN prior tables exactly as librosa builds them per call (core/pitch.py::pyin:
        scipy.stats.beta.cdf over linspace thresholds, scipy.stats.boltzmann.pmf, the pitch-bin
        frequencies), so the kernels work from the same float64 values as the reference."""
        import scipy.stats
        thresholds = np.linspace(0, 1, 101)
        self.set_table("beta_probs", np.diff(scipy.stats.beta.cdf(thresholds, 2, 18)))
        n = len(self.table("boltz_fact"))
        lam, idx = 2.0, np.arange(n)
        with np.errstate(divide="ignore", invalid="ignore"):
            fact = (1 - np.exp(-lam)) / (1 - np.exp(-lam * idx))      # scipy.stats.boltzmann._pmf
        fact[0] = 0.0
        self.set_table("boltz_fact", fact)
        self.set_table("boltz_exp", np.exp(-lam * idx))
        fmin, nb = self.table("freqs")[0], self.param("n_pitch_bins")
        self.set_table("freqs", fmin * 2 ** (np.arange(nb) / 120))

    def set_table(self, name, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        self._check(self.lib.aegis_set_table(self._h, name.encode(), v.ctypes.data, len(v)))

    def close(self):
        if getattr(self, "_h", None):
            h, self._h = self._h, None
            self.lib.aegis_destroy(h)

    def __del__(self):
        # from the garbage collector: never during interpreter finalisation (the atexit hook above has closed every live
        # handle by then; a handle created later than that is left to the process exit)
        if not sys.is_finalizing():
            self.close()

    def _check(self, rc):
        if rc != OK:
            raise AegisError(rc, self.lib.aegis_last_error(self._h).decode())

    def frames_for(self, n_samples):
        return int(self.lib.aegis_frames_for(self._h, int(n_samples)))

    def param(self, name):
        v = int(self.lib.aegis_get_param(self._h, name.encode()))
        if v < 0:
            raise AegisError(v, f"unknown parameter {name}")
        return v

    def table(self, name):
        n = int(self.lib.aegis_get_table(self._h, name.encode(), None, 0))
        if n < 0:
            raise AegisError(n, f"unknown table {name}")
        out = np.empty(n, dtype=_TABLE_DTYPES.get(name, np.float64))
        self.lib.aegis_get_table(self._h, name.encode(), out.ctypes.data, n)
        return out

    def debug_fetch(self, name):
        n = int(self.lib.aegis_debug_fetch(self._h, name.encode(), None, 0))
        if n < 0:
            raise AegisError(n, self.lib.aegis_last_error(self._h).decode())
        out = np.empty(n, dtype=_DEBUG_DTYPES.get(name, np.float64))
        got = int(self.lib.aegis_debug_fetch(self._h, name.encode(), out.ctypes.data, n))
        if got < 0:
            raise AegisError(got, self.lib.aegis_last_error(self._h).decode())
        return out

    def set_profiling(self, on=True):
        self._check(self.lib.aegis_set_profiling(self._h, 1 if on else 0))

    def kernel_ms(self, name):
        return float(self.lib.aegis_last_kernel_ms(self._h, name.encode()))

    def kernel_launches(self, name):
        return int(self.lib.aegis_last_kernel_launches(self._h, name.encode()))

    def viterbi_stats(self, reset=True):
        """{"wave_steps", "list_only", "skipped"}: wave-steps of the band Viterbi since the last reset, how many of them
        took the exact observed-sources-only path, and how many were voiced waves with nothing but dead targets at an
        easy frame, which skip the step (viterbi.hip); None on a host-only handle."""
        if self.device < 0:
            return None
        v = np.zeros(3, np.int64)
        n = int(self.lib.aegis_debug_fetch(self._h, b"viterbi_stats" if reset else b"viterbi_stats_peek", v.ctypes.data, 3))
        if n < 0:
            return None
        return {"wave_steps": int(v[0]), "list_only": int(v[1]), "skipped": int(v[2])}

    def analyze_batch(self, clips, rake_sensitivity=0.6, stages=STAGE_ALL, want_sdb=True, check_finite=False,
                      f0_zero=False, views=False, concatenated=False, want_col_means=False):
        """clips: list of float32 1-D arrays (host).  Returns a list of per-clip dicts with the
        dtypes of the reference's raw_data (aegis_engine.py:72-75); f0 keeps NaN where unvoiced unless f0_zero
        (np.nan_to_num, aegis_engine.py:69).  check_finite: librosa's valid_audio test on the device -> ValueError.
        views: the per-clip arrays are slices of the batch's buffers instead of copies.  concatenated: also return
        (buffers dict, frame offsets) of the whole batch, for the batched event extraction."""
        clips = [np.ascontiguousarray(c, dtype=np.float32) for c in clips]
        n = len(clips)
        if n == 0:
            return ([], {}, np.zeros(1, np.int64)) if concatenated else []
        ptrs = (C.c_void_p * n)(*[c.ctypes.data for c in clips])
        lens = (C.c_int64 * n)(*[len(c) for c in clips])
        frames = [1 + len(c) // self.hop for c in clips]
        F = sum(frames)
        if stages & STAGE_RAKE:
            stages |= STAGE_MEL
        bufs = {}
        out = Outputs()
        if stages & STAGE_PYIN:
            bufs["f0"] = np.empty(F, np.float64)
            bufs["voiced_flag"] = np.empty(F, np.uint8)
            bufs["voiced_prob"] = np.empty(F, np.float64)
            if concatenated:
                bufs["pitch_bin"] = np.empty(F, np.int16)     # batch-level only: not part of the per-clip dicts
        if stages & STAGE_RMS:
            bufs["rms"] = np.empty(F, np.float32)
        if stages & STAGE_RAKE:
            bufs["rake_mask"] = np.empty(F, np.uint8)
        if (stages & STAGE_MEL) and want_sdb:
            bufs["S_dB"] = np.empty(F * self.n_mels, np.float32)
        if (stages & STAGE_MEL) and want_col_means:
            bufs["sdb_col_means"] = np.empty(3 * F, np.float32)     # batch-level only: [3][F] all / low half / high half
        for k, v in bufs.items():
            setattr(out, k, v.ctypes.data)
        flags = int(stages) | (OPT_CHECK_FINITE if check_finite else 0) | (OPT_F0_ZERO if f0_zero else 0)
        rc = self.lib.aegis_analyze_batch(self._h, ptrs, lens, n, float(rake_sensitivity), flags, C.byref(out))
        if rc == ERR_INVALID and check_finite:
            msg = self.lib.aegis_last_error(self._h).decode()
            if msg.startswith("Audio buffer is not finite"):
                raise ValueError(msg)                       # librosa.util.valid_audio's ParameterError
        self._check(rc)
        for k in ("voiced_flag", "rake_mask"):              # 0 / 1 bytes: the same memory read as bool
            if k in bufs:
                bufs[k] = bufs[k].view(bool)
        res, fo = [], 0
        for Fc in frames:
            d = {}
            for k, v in bufs.items():
                if k in ("pitch_bin", "sdb_col_means"):
                    continue
                if k == "S_dB":
                    a = v[fo * self.n_mels:(fo + Fc) * self.n_mels].reshape(self.n_mels, Fc)
                else:
                    a = v[fo:fo + Fc]
                d[k] = a if views else a.copy()
            res.append(d)
            fo += Fc
        if concatenated:
            return res, bufs, np.concatenate([[0], np.cumsum(frames)]).astype(np.int64)
        return res

    def rake_patterns(self, S_dB, ratio):
        S = np.ascontiguousarray(S_dB, dtype=np.float32)
        n_mels, F = S.shape
        out = np.zeros(F, np.uint8)
        self._check(self.lib.aegis_rake_patterns(self._h, S.ctypes.data, n_mels, F, float(ratio), out.ctypes.data))
        return out.astype(bool)

    def trend(self, op, series, params, n_out=1, out_dtype=np.float64, stacked_rows=None):
        """aegis_trend over a list of float64 series (or one `stacked_rows` x len array for the
        consensus op).  Returns a list (per output) of lists (per series) of arrays."""
        if stacked_rows is not None:
            x = np.ascontiguousarray(series, dtype=np.float64).reshape(stacked_rows, -1)
            lens = [x.shape[1]]
            flat = x.ravel()
        else:
            series = [np.ascontiguousarray(s, dtype=np.float64) for s in series]
            lens = [len(s) for s in series]
            flat = np.concatenate(series) if series else np.zeros(0)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        total = int(off[-1])
        par = np.ascontiguousarray(params, dtype=np.float64)
        dtypes = out_dtype if isinstance(out_dtype, (list, tuple)) else [out_dtype] * n_out
        outs = [np.empty(total, dtype=dt) for dt in dtypes]
        ptrs = (C.c_void_p * n_out)(*[o.ctypes.data for o in outs])
        self._check(self.lib.aegis_trend(self._h, int(op), flat.ctypes.data, off.ctypes.data, len(lens),
                                         par.ctypes.data, len(par), ptrs, n_out))
        return [[o[off[i]:off[i + 1]] for i in range(len(lens))] for o in outs]

    def ghost_rsi(self, ev_a, ev_b, event_off, track_len, period=14):
        """aegis_ghost_rsi: the Wilder averages (avg_gain, avg_loss) of every clip's ghost-note density track at its notes'
        own positions; ev_a / ev_b = int(start*10) / int(end*10) of the notes clip after clip, event_off their offsets."""
        a = np.ascontiguousarray(ev_a, dtype=np.int64)
        b = np.ascontiguousarray(ev_b, dtype=np.int64)
        off = np.ascontiguousarray(event_off, dtype=np.int64)
        tl = np.ascontiguousarray(track_len, dtype=np.int64)
        g, l = np.full(len(a), np.nan), np.full(len(a), np.nan)
        self._check(self.lib.aegis_ghost_rsi(self._h, a.ctypes.data, b.ctypes.data, off.ctypes.data, len(tl), tl.ctypes.data,
                                             int(period), g.ctypes.data, l.ctypes.data))
        return g, l

    def cqt(self, clips, n_bins=84, bins_per_octave=12, fmin=32.70319566257483, filter_scale=1.0):
        """|CQT| of each clip, float32 [n_bins, 1 + len//hop] (aegis_cqt: direct transform on the MFMA units)."""
        clips = [np.ascontiguousarray(c, dtype=np.float32) for c in clips]
        n = len(clips)
        if n == 0:
            return []
        ptrs = (C.c_void_p * n)(*[c.ctypes.data for c in clips])
        lens = (C.c_int64 * n)(*[len(c) for c in clips])
        frames = [self.frames_for(len(c)) for c in clips]
        out = np.empty(sum(frames) * n_bins, np.float32)
        self._check(self.lib.aegis_cqt(self._h, ptrs, lens, n, n_bins, bins_per_octave, float(fmin), float(filter_scale),
                                       out.ctypes.data))
        res, o = [], 0
        for Fc in frames:
            res.append(out[o:o + Fc * n_bins].reshape(n_bins, Fc).copy())
            o += Fc * n_bins
        return res

    def chroma_cqt(self, clips, bin_class, n_chroma=12, n_bins=252, bins_per_octave=36, fmin=32.70319566257483, filter_scale=1.0):
        """aegis_chroma_cqt: |CQT| folded into chroma classes (bin_class[n_bins] -> 0..n_chroma-1) and divided by the frame
        maximum, all on the device: float32 [n_chroma, 1 + len//hop] per clip."""
        clips = [np.ascontiguousarray(c, dtype=np.float32) for c in clips]
        n = len(clips)
        if n == 0:
            return []
        ptrs = (C.c_void_p * n)(*[c.ctypes.data for c in clips])
        lens = (C.c_int64 * n)(*[len(c) for c in clips])
        frames = [self.frames_for(len(c)) for c in clips]
        cls = np.ascontiguousarray(bin_class, dtype=np.int32)
        if len(cls) != n_bins:
            raise ValueError("bin_class needs one entry per CQT bin")
        out = np.empty(sum(frames) * n_chroma, np.float32)
        self._check(self.lib.aegis_chroma_cqt(self._h, ptrs, lens, n, n_bins, bins_per_octave, float(fmin), float(filter_scale),
                                              n_chroma, cls.ctypes.data, out.ctypes.data))
        res, o = [], 0
        for Fc in frames:
            res.append(out[o:o + Fc * n_chroma].reshape(n_chroma, Fc).copy())
            o += Fc * n_chroma
        return res

    def cqt_device(self, d_pcm_ptr, sample_offsets, d_out_ptr, n_bins=84, bins_per_octave=12, fmin=32.70319566257483,
                   filter_scale=1.0, stream=None, sync=True):
        """aegis_cqt_device: PCM and |CQT| resident in device memory (raw pointers as ints)."""
        off = np.ascontiguousarray(sample_offsets, dtype=np.int64)
        self._check(self.lib.aegis_cqt_device(self._h, C.c_void_p(int(d_pcm_ptr)), off.ctypes.data_as(C.POINTER(C.c_int64)),
                                              len(off) - 1, n_bins, bins_per_octave, float(fmin), float(filter_scale),
                                              C.c_void_p(int(d_out_ptr)), C.c_void_p(stream or 0), 1 if sync else 0))

    def open_stream(self, max_seconds=600.0):
        return Stream(self, int(max_seconds * self.sr))

    def analyze_batch_device(self, d_pcm_ptr, sample_offsets, outputs, rake_sensitivity=0.6,
                             stages=STAGE_ALL, stream=None, sync=True):
        """PCM and outputs already in device memory (raw pointers as ints).  `outputs` maps the
        aegis_outputs field names to device pointers."""
        off = np.ascontiguousarray(sample_offsets, dtype=np.int64)
        out = Outputs()
        for k, v in outputs.items():
            setattr(out, k, int(v) if v else None)
        self._check(self.lib.aegis_analyze_batch_device(
            self._h, C.c_void_p(int(d_pcm_ptr)), off.ctypes.data_as(C.POINTER(C.c_int64)), len(off) - 1,
            float(rake_sensitivity), int(stages), C.byref(out), C.c_void_p(stream or 0), 1 if sync else 0))


class Stream:
    """Incremental analysis of one clip (aegis_stream_*).  push() returns the frames that became complete:
    dict(rms, voiced_prob, live_state); close() returns the same dict analyze_batch() gives for the whole
    signal (bit-identical)."""

    def __init__(self, handle, max_samples):
        self.handle = handle
        self.lib = handle.lib
        s = C.c_void_p()
        handle._check(self.lib.aegis_stream_open(handle._h, int(max_samples), C.byref(s)))
        self._s = s
        self._cap = 8 + max_samples // handle.hop
        self._rms = np.empty(self._cap, np.float32)
        self._vp = np.empty(self._cap, np.float64)
        self._live = np.empty(self._cap, np.int32)
        self._frames = StreamFrames(self._rms.ctypes.data, self._vp.ctypes.data, self._live.ctypes.data)
        self.n_bins = handle.param("n_pitch_bins")

    def push(self, samples):
        x = np.ascontiguousarray(samples, dtype=np.float32)
        k = C.c_int64(0)
        self.handle._check(self.lib.aegis_stream_push(self._s, x.ctypes.data, len(x), C.byref(self._frames), C.byref(k)))
        n = k.value
        return {"rms": self._rms[:n].copy(), "voiced_prob": self._vp[:n].copy(), "live_state": self._live[:n].copy()}

    def close(self, rake_sensitivity=0.6, want_sdb=True):
        h = self.handle
        cap = self._cap
        bufs = {"f0": np.empty(cap, np.float64), "voiced_flag": np.empty(cap, np.uint8), "voiced_prob": np.empty(cap, np.float64),
                "rms": np.empty(cap, np.float32), "rake_mask": np.empty(cap, np.uint8)}
        if want_sdb:
            bufs["S_dB"] = np.empty(cap * h.n_mels, np.float32)
        out = Outputs()
        for k, v in bufs.items():
            setattr(out, k, v.ctypes.data)
        F = C.c_int64(0)
        h._check(self.lib.aegis_stream_close(self._s, float(rake_sensitivity), C.byref(out), C.byref(F)))
        F = F.value
        res = {}
        for k, v in bufs.items():
            if k == "S_dB":
                res[k] = v[:F * h.n_mels].reshape(h.n_mels, F).copy()
            elif k in ("voiced_flag", "rake_mask"):
                res[k] = v[:F].astype(bool)
            else:
                res[k] = v[:F].copy()
        return res

    def free(self):
        # safe in either order with Handle.close(): the C handle outlives its open streams (aegis_destroy defers)
        if getattr(self, "_s", None):
            self.lib.aegis_stream_free(self._s)
            self._s = None

    def __del__(self):
        if not sys.is_finalizing():
            self.free()
