"""`AegisEngine`: the reference's engine facade (/root/reference/aegis_engine.py:16-216) over
the MI355X kernels.  Same constructor, method names, keyword arguments, return types and error
behaviour, so the Streamlit/FastAPI callers and midi_logic consume it unchanged:

    engine = AegisEngine()                       # aegis_engine.py:17
    raw = engine.audio_to_midi(wav, None, turbo_mode=True, rake_sensitivity=0.6)   # :41-75
    events = engine.extract_events(raw, "out.mid", confidence_threshold=0.7)        # :77-181

The per-frame arithmetic (mel/dB/rake, pYIN, RMS) runs in libaegis_hip.so; there is no CPU
path -- without the extension or a GPU every analyze call raises.
"""
import multiprocessing

import numpy as np

from . import _lib, audio_io, events_native, midi_logic, smf
from .convert import note_to_hz

_FMIN, _FMAX = note_to_hz("E2"), note_to_hz("C6")
_ENGINE_ONLY_KWARGS = ("confidence_threshold", "start_time", "end_time", "turbo_mode", "rake_sensitivity",
                       "vibrato_rate", "vibrato_depth")


def _require_finite(y):
    # librosa.util.valid_audio raises ParameterError on NaN / inf samples.  A BLAS dot product is non-finite whenever
    # a sample is (NaN and inf propagate) and costs a quarter of np.isfinite(y).all(); only if it is non-finite --
    # which finite samples beyond 1e19 could also cause -- is the exact test run.
    if not np.isfinite(np.dot(y, y)) and not np.isfinite(y).all():
        raise ValueError("Audio buffer is not finite everywhere")


class AegisEngine:
    def __init__(self, sample_rate=44100, hop_length=512, n_fft=2048, device=0, verbose=False, pyin_init="unvoiced"):
        # pyin_init (not a reference argument): "unvoiced" = librosa.pyin's own initial distribution (the default, what
        # the reference gets from librosa), "uniform" = the alternative reading documented in DESIGN.md section 1
        self.pyin_init = pyin_init
        self.sr = sample_rate
        self.hop_length = hop_length
        self.n_fft = n_fft
        self.device = device
        self.verbose = verbose
        self.turbo_cores = None      # None -> multiprocessing.cpu_count(), as aegis_engine.py:187
        self._handle = None
        self._freqs = None

    # ------------------------------------------------------------------ device context
    @property
    def handle(self):
        if self._handle is None:
            self._handle = _lib.Handle(sample_rate=self.sr, hop_length=self.hop_length, n_fft=self.n_fft,
                                       n_mels=128, fmin=_FMIN, fmax=_FMAX, device=self.device, pyin_init=self.pyin_init)
        return self._handle

    def close(self):
        if self._handle is not None:
            self._handle.close()
            self._handle = None

    def _say(self, msg):
        if self.verbose:
            print(msg)

    # ------------------------------------------------------------------ reference surface
    def load_audio(self, file_path, start_time=0, end_time=None):
        """-> (y float32[N], S_dB float32[128, F])  (aegis_engine.py:22-27)."""
        duration = (end_time - start_time) if end_time else None
        y = audio_io.read_wav(file_path, self.sr, offset=start_time, duration=duration)
        _require_finite(y)
        out = self.handle.analyze_batch([y], stages=_lib.STAGE_MEL)[0]
        return y, out["S_dB"]

    def detect_rake_patterns(self, S_dB):
        """aegis_engine.py:38-39 (fixed 0.6 ratio)."""
        return self.handle.rake_patterns(S_dB, 0.6)

    # Pass-throughs (aegis_engine.py:29-36).  Stem separation, tab generation and MusicXML export consume file paths /
    # the event list unchanged and are outside the MI355X path: when the reference's own package is importable (the
    # drop-in case: this class swapped into the reference tree, INTEGRATION.md option A) the calls are forwarded to
    # it exactly as the reference forwards them; otherwise they raise.
    @staticmethod
    def _reference_core(module, name):
        import importlib
        try:
            return getattr(importlib.import_module(f"aegis_engine_core.{module}"), name)
        except ImportError as e:
            raise NotImplementedError(
                f"{name} forwards to the reference's aegis_engine_core.{module} (aegis_engine.py:29-36), which is not "
                f"importable here ({e}); it is outside the MI355X analyze path") from e

    def separate_stems(self, input_wav, output_dir):
        return self._reference_core("stems", "separate_stems")(input_wav, output_dir)

    def generate_tabs(self, events):
        return self._reference_core("tabs", "generate_tabs")(events)

    def export_musicxml(self, tab_data, xml_path):
        return self._reference_core("tabs", "export_musicxml")(tab_data, xml_path)

    def audio_to_midi(self, input_wav, output_mid, **kwargs):
        """Perception phase -> raw_data dict or None for empty audio (aegis_engine.py:41-75).
        `output_mid` is accepted and ignored, as in the reference."""
        start_time, end_time = kwargs.get("start_time", 0), kwargs.get("end_time", None)
        duration = (end_time - start_time) if end_time else None
        y = audio_io.read_wav(input_wav, self.sr, offset=start_time, duration=duration)
        return self.analyze_array(y, turbo_mode=kwargs.get("turbo_mode", False),
                                  rake_sensitivity=kwargs.get("rake_sensitivity", 0.6))

    analyze = audio_to_midi      # BASELINE.json's name for the same call

    def analyze_array(self, y, turbo_mode=False, rake_sensitivity=0.6):
        """audio_to_midi from decoded PCM onward (aegis_engine.py:51-75)."""
        res = self.analyze_arrays([y], turbo_mode=turbo_mode, rake_sensitivity=rake_sensitivity)
        return res[0]

    def analyze_arrays(self, clips, turbo_mode=False, rake_sensitivity=0.6, _concatenated=False):
        """Batch form: one ragged GPU batch for a folder of clips; element i is what
        audio_to_midi returns for clip i (None for an empty clip).  The arrays of a batch are slices of the batch's
        buffers (one allocation per output, not one per clip)."""
        clips = [np.ascontiguousarray(c, dtype=np.float32) for c in clips]
        live = [i for i, c in enumerate(clips) if len(c) > 0]
        results = [None] * len(clips)
        if not live:
            return (results, None, None, live) if _concatenated else results
        self._say(f"[Aegis] Starting Perception Phase (Turbo: {turbo_mode})...")
        h = self.handle
        # librosa.util.valid_audio (NaN / infinite samples raise) is tested on the device, where the samples are read anyway
        bufs = off = None
        if turbo_mode:
            frames = h.analyze_batch([clips[i] for i in live], rake_sensitivity=rake_sensitivity, check_finite=True,
                                     stages=_lib.STAGE_MEL | _lib.STAGE_RAKE | _lib.STAGE_RMS, want_sdb=False)
            for i, r in zip(live, frames):
                try:
                    r["f0"], r["voiced_flag"], r["voiced_prob"] = self._parallel_pitch_tracking(clips[i])
                except _lib.AegisError:
                    raise
                except Exception as e:    # aegis_engine.py:61-63: fall back to the stable path
                    self._say(f"[Aegis] Parallel failed ({e}), falling back to stable core.")
                    p = h.analyze_batch([clips[i]], stages=_lib.STAGE_PYIN)[0]
                    r["f0"], r["voiced_flag"], r["voiced_prob"] = p["f0"], p["voiced_flag"], p["voiced_prob"]
                r["f0"] = np.nan_to_num(r["f0"])
        else:
            self._say("[Aegis] Using Stable Single-core Analysis.")
            frames, bufs, off = h.analyze_batch([clips[i] for i in live], rake_sensitivity=rake_sensitivity,
                                                stages=_lib.STAGE_ALL, want_sdb=False, check_finite=True, f0_zero=True,
                                                views=True, concatenated=True)
        for i, r in zip(live, frames):
            results[i] = {"rake_mask": r["rake_mask"], "f0": r["f0"],
                          "voiced_flag": r["voiced_flag"], "voiced_probs": r["voiced_prob"],
                          "rms": r["rms"], "y": clips[i]}
        return (results, bufs, off, live) if _concatenated else results

    def audio_to_midi_batch(self, clips, want_midi=True, **kwargs):
        """A folder of decoded clips -> (raw_data dicts, event lists, SMF bytes per clip): audio_to_midi + extract_events
        for every clip (aegis_engine.py:41-181) as ONE analysis batch and ONE batched event extraction / MIDI rendering
        (C++, clips in parallel).  Empty clips give (None, [], None).  kwargs as for both reference methods."""
        turbo = kwargs.get("turbo_mode", False)
        raws, bufs, off, live = self.analyze_arrays(clips, turbo_mode=turbo, rake_sensitivity=kwargs.get("rake_sensitivity", 0.6),
                                                    _concatenated=True)
        events, blobs = [[] for _ in clips], [None] * len(clips)
        if not live:
            return raws, events, blobs
        passthrough = {k: v for k, v in kwargs.items() if k not in _ENGINE_ONLY_KWARGS + ("midi_program",)}
        smf_kw = dict(midi_program=kwargs.get("midi_program", 27), vibrato_rate=kwargs.get("vibrato_rate", 5.0),
                      vibrato_depth=kwargs.get("vibrato_depth", 0.3))
        if bufs is None:         # Turbo Mode: per-clip arrays of different lengths, truncated as extract_events does
            cat = {k: [] for k in ("rake_mask", "f0", "voiced_flag", "voiced_probs", "rms")}
            lens = []
            for i in live:
                r = raws[i]
                n = min(len(r["rake_mask"]), len(r["f0"]), len(r["rms"]))
                lens.append(n)
                for k in cat:
                    cat[k].append(r[k][:n])
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            bufs = {("voiced_prob" if k == "voiced_probs" else k): np.concatenate(v) for k, v in cat.items()}
        grid = {}
        if "pitch_bin" in bufs:       # the analysis's own bins: hz_to_midi from a 441-entry table
            if self._freqs is None:
                self._freqs = self.handle.table("freqs")
            grid = dict(pitch_bin=bufs["pitch_bin"], freqs=self._freqs)
        res = events_native.extract_batch(off, bufs["rake_mask"], bufs["f0"], bufs["voiced_flag"], bufs["voiced_prob"], bufs["rms"],
                                          self.sr, self.hop_length, kwargs.get("confidence_threshold", 0.70),
                                          want_midi=want_midi, **grid, **smf_kw, **passthrough)
        per, bl = res if want_midi else (res, None)
        for j, i in enumerate(live):
            events[i] = per[j]
            if want_midi:
                blobs[i] = bl[j]
        return raws, events, blobs

    def extract_events(self, raw_data, output_mid, **kwargs):
        """Logic filter layer (aegis_engine.py:77-181): raw_data -> events, optional SMF to a
        path or a file-like object."""
        keys = ("rake_mask", "f0", "voiced_flag", "voiced_probs", "rms")
        n = min(len(raw_data["rake_mask"]), len(raw_data["f0"]), len(raw_data["rms"]))
        rake_mask, f0, voiced_flag, voiced_probs, rms = (raw_data[k][:n] for k in keys)
        passthrough = {k: v for k, v in kwargs.items() if k not in _ENGINE_ONLY_KWARGS + ("midi_program",)}
        smf_kw = dict(midi_program=kwargs.get("midi_program", 27), vibrato_rate=kwargs.get("vibrato_rate", 5.0),
                      vibrato_depth=kwargs.get("vibrato_depth", 0.3))
        res = events_native.extract_batch([0, n], rake_mask, f0, voiced_flag, voiced_probs, rms, self.sr, self.hop_length,
                                          kwargs.get("confidence_threshold", 0.70), want_midi=bool(output_mid), **smf_kw,
                                          **passthrough)
        if output_mid:
            events, blob = res[0][0], res[1][0]
            if hasattr(output_mid, "write"):
                output_mid.write(blob)
            else:
                with open(output_mid, "wb") as f:
                    f.write(blob)
        else:
            events = res[0]
        return events

    # ------------------------------------------------------------------ Turbo Mode
    def _turbo_spans(self, n_samples):
        """Equal frame spans per core, last one takes the remainder (aegis_engine.py:192-204)."""
        cores = self.turbo_cores or multiprocessing.cpu_count()
        total = int(np.ceil(n_samples / self.hop_length))
        per = total // cores or total
        spans = []
        for i in range(cores):
            lo = i * per
            if lo >= total:
                break
            hi = (i + 1) * per if i < cores - 1 else total
            a, b = lo * self.hop_length, min(hi * self.hop_length, n_samples)
            if b > a:
                spans.append((a, b))
        return spans

    def _parallel_pitch_tracking(self, y):
        """Turbo Mode (aegis_engine.py:183-216): the clip is cut into cpu_count() time chunks,
        each chunk is pYIN-tracked on its own (own centring, own Viterbi) and the results are
        concatenated.  The reference maps chunks over a process pool; here they are the clips of
        one GPU batch, so every chunk's Viterbi runs on its own compute unit concurrently."""
        y = np.ascontiguousarray(y, dtype=np.float32)
        h = self.handle
        if len(y) / self.sr < 5.0:
            r = h.analyze_batch([y], stages=_lib.STAGE_PYIN)[0]
            return r["f0"], r["voiced_flag"], r["voiced_prob"]
        parts = h.analyze_batch([y[a:b] for a, b in self._turbo_spans(len(y))], stages=_lib.STAGE_PYIN)
        return (np.concatenate([p["f0"] for p in parts]), np.concatenate([p["voiced_flag"] for p in parts]),
                np.concatenate([p["voiced_prob"] for p in parts]))
