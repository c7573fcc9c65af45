"""`AegisFinancialEngine`: the reference's v2 engine facade (/root/reference/aegis_engine_financial.py:23-253,
default rate 22 050 Hz) over the MI355X kernels: load -> mel/dB -> rake -> pYIN -> guitar filters -> RMS ->
financial event extraction -> two-track SMF.  Same constructor, methods, kwargs and return values."""
import struct

import numpy as np

from . import _lib, audio_io
from .convert import note_to_hz
from .guitar import apply_guitar_filters, apply_guitar_filters_from_means
from .midi_logic_financial import get_midi_events_financial, get_midi_events_financial_batch
from .smf import Track


class AegisFinancialEngine:
    def __init__(self, sample_rate=22050, hop_length=512, n_fft=2048, device=0, verbose=False, pyin_init="unvoiced"):
        self.pyin_init = pyin_init      # see AegisEngine
        self.sr = sample_rate
        self.hop_length = hop_length
        self.n_fft = n_fft
        self.version = "2.0-Financial"
        self.device = device
        self.verbose = verbose
        self._handle = None

    @property
    def handle(self):
        if self._handle is None:
            self._handle = _lib.Handle(sample_rate=self.sr, hop_length=self.hop_length, n_fft=self.n_fft,
                                       fmin=note_to_hz("E2"), fmax=note_to_hz("C6"), device=self.device,
                                       pyin_init=self.pyin_init)
        return self._handle

    def load_audio(self, file_path, start_time=0, end_time=None):
        duration = (end_time - start_time) if end_time else None
        y = audio_io.read_wav(file_path, self.sr, offset=start_time, duration=duration)
        return y, self.handle.analyze_batch([y], stages=_lib.STAGE_MEL)[0]["S_dB"]

    def detect_rake_patterns(self, S_dB, rake_sensitivity=0.6):
        return self.handle.rake_patterns(S_dB, rake_sensitivity)

    def pitch_tracking(self, y):
        r = self.handle.analyze_batch([np.asarray(y, np.float32)], stages=_lib.STAGE_PYIN)[0]
        return r["f0"], r["voiced_flag"], r["voiced_prob"]

    def analyze_array(self, y, **kwargs):
        """Everything of audio_to_midi_financial up to the event list, from decoded PCM (one GPU call)."""
        y = np.ascontiguousarray(y, dtype=np.float32)
        r = self.handle.analyze_batch([y], rake_sensitivity=kwargs.get("rake_sensitivity", 0.6))[0]
        f0, voiced, rake = r["f0"], r["voiced_flag"], r["rake_mask"]
        if kwargs.get("use_guitar_filters", True):
            g = apply_guitar_filters(f0, voiced, r["S_dB"], self.hop_length, self.sr, rake)
            f0, rake = g["f0"], g["rake_mask"]
            voiced = g["voiced"] & ~g["mute_mask"]
        passthrough = {k: v for k, v in kwargs.items()
                       if k not in ("confidence_threshold", "rake_sensitivity", "use_financial")}
        return get_midi_events_financial(
            rake_mask=rake, f0=f0, voiced_flag=voiced, active_probs=r["voiced_prob"], rms=r["rms"], sr=self.sr,
            hop_length=self.hop_length, confidence_threshold=kwargs.get("confidence_threshold", None),
            use_financial=kwargs.get("use_financial", True), verbose=self.verbose, **passthrough)

    def analyze_arrays(self, clips, **kwargs):
        """analyze_array for a folder of decoded clips: ONE ragged GPU analysis batch (the guitar filters read the three
        column means of the dB image the library computes beside it, so the image itself never leaves the GPU), ONE fused
        pitch-analysis call and ONE ghost-note RSI call for all clips (midi_logic_financial.get_midi_events_financial_batch).
        Element i is what analyze_array(clips[i], **kwargs) returns.  A clip too short for the trend filters (fewer than 10
        frames: analyze_array raises IndexError, as the reference does) stays out of the fused calls: with
        return_exceptions=True its element is that exception, otherwise the exception is raised after all other clips have
        been processed, the full result list attached as `.results`."""
        clips = [np.ascontiguousarray(c, dtype=np.float32) for c in clips]
        if not clips:
            return []
        use_guitar = kwargs.get("use_guitar_filters", True)
        res, bufs, off = self.handle.analyze_batch(clips, rake_sensitivity=kwargs.get("rake_sensitivity", 0.6), want_sdb=False,
                                                   want_col_means=use_guitar, views=True, concatenated=True)
        F = int(off[-1])
        items = []
        for i, r in enumerate(res):
            f0, voiced, rake = r["f0"], r["voiced_flag"], r["rake_mask"]
            if use_guitar:
                a, b = int(off[i]), int(off[i + 1])
                cm = bufs["sdb_col_means"]
                g = apply_guitar_filters_from_means(f0, voiced, (cm[a:b], cm[F + a:F + b], cm[2 * F + a:2 * F + b]),
                                                    self.hop_length, self.sr, rake)
                f0, rake = g["f0"], g["rake_mask"]
                voiced = g["voiced"] & ~g["mute_mask"]
            items.append({"rake_mask": rake, "f0": f0, "voiced_flag": voiced, "active_probs": r["voiced_prob"], "rms": r["rms"]})
        passthrough = {k: v for k, v in kwargs.items()
                       if k not in ("confidence_threshold", "rake_sensitivity", "use_financial", "return_exceptions")}
        return get_midi_events_financial_batch(items, self.sr, self.hop_length,
                                               confidence_threshold=kwargs.get("confidence_threshold", None),
                                               use_financial=kwargs.get("use_financial", True), verbose=self.verbose,
                                               return_exceptions=kwargs.get("return_exceptions", False), **passthrough)

    def render_midi(self, events):
        """aegis_engine_financial.py:190-245: track_name metas, note_on at the start tick, note_off after
        the duration, per-track running clock (type 1, 480 ticks per beat, 120 BPM tick length)."""
        ms_per_tick = 500 / 480
        ms_per_frame = (self.hop_length / self.sr) * 1000
        tracks = {"main": Track(), "safe": Track()}
        blobs = {}
        for name, label in (("main", b"Aegis Financial - Main"), ("safe", b"Aegis Financial - Safe")):
            blobs[name] = b"\x00\xff\x03" + bytes([len(label)]) + label
        for e in events:
            t = tracks["main" if e["track"] == "main" else "safe"]
            start = int(e["start"] * ms_per_frame / ms_per_tick)
            length = int((e["end"] - e["start"]) * ms_per_frame / ms_per_tick)
            t.note_on(start, e["note"], e["velocity"])
            t.note_off(start + length, e["note"], 0)
        out = b"MThd" + struct.pack(">IHHH", 6, 1, 2, 480)
        for name in ("main", "safe"):
            body = blobs[name] + bytes(tracks[name]._data) + b"\x00\xff\x2f\x00"
            out += b"MTrk" + struct.pack(">I", len(body)) + body
        return out

    def audio_to_midi_financial(self, input_wav, output_mid, **kwargs):
        """-> output_mid, or None when no note was found (aegis_engine_financial.py:73-253)."""
        y = audio_io.read_wav(input_wav, self.sr)
        events = self.analyze_array(y, **kwargs)
        if not events:
            return None
        with open(output_mid, "wb") as f:
            f.write(self.render_midi(events))
        return output_mid
