"""v2 financial event extraction end to end on the GPU trend filters, against events produced by the
reference's own midi_logic_financial.py on the same frame arrays -- PINNED (librosa one-liners stubbed)."""
import json
import os

import numpy as np
import pytest

from oracle import smf as osmf
from spectrogram_midi_amd import audio_io
from tools import signals
from spectrogram_midi_amd.engine_financial import AegisFinancialEngine
from spectrogram_midi_amd.guitar import apply_guitar_filters
from spectrogram_midi_amd.midi_logic_financial import get_midi_events_financial

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(__file__)
G = np.load(os.path.join(HERE, "golden", "v2_engine_golden.npz"))
META = json.load(open(os.path.join(HERE, "golden", "v2_engine_golden.json")))
KW = {"default": {}, "fixed_thr": {"confidence_threshold": 0.6, "min_note_duration_ms": 80},
      "no_harm": {"use_harmonic_filter": False, "sustain_ms": 120}, "legacy": {"use_financial": False}}


def test_financial_events_match_reference():
    for name, m in META.items():
        g = apply_guitar_filters(G[f"{name}/f0"], G[f"{name}/voiced"], G[f"{name}/S_dB"], 512, 22050, G[f"{name}/rake"])
        voiced = g["voiced"] & ~g["mute_mask"]
        for tag, kw in KW.items():
            ev = get_midi_events_financial(rake_mask=g["rake_mask"], f0=g["f0"], voiced_flag=voiced,
                                           active_probs=G[f"{name}/vprob"], rms=G[f"{name}/rms"], sr=22050,
                                           hop_length=512, **kw)
            ref = m["events"][tag]
            assert len(ev) == len(ref), (name, tag)
            for a, b in zip(ev, ref):
                assert set(a) == set(b), (name, tag, set(a) ^ set(b))
                for k in b:
                    if k == "confidence":
                        assert abs(float(a[k]) - b[k]) <= 1e-9, (name, tag, k)
                    elif k == "key_info":
                        assert (a[k]["key"], a[k]["mode"]) == (b[k]["key"], b[k]["mode"])
                    else:
                        assert a[k] == b[k], (name, tag, k, a[k], b[k])


def test_engine_end_to_end(tmp_path):
    y = signals.guitar_clip(10.0, sr=22050, seed=21)
    wav, mid = str(tmp_path / "in.wav"), str(tmp_path / "out.mid")
    audio_io.write_wav(wav, y, 22050)
    eng = AegisFinancialEngine()
    assert eng.version == "2.0-Financial" and eng.sr == 22050
    out = eng.audio_to_midi_financial(wav, mid)
    assert out == mid
    typ, tpb, tracks = osmf.parse_smf(open(mid, "rb").read())
    assert (typ, tpb, len(tracks)) == (1, 480, 2)
    assert tracks[0][0][1] == 0xFF and tracks[0][0][2][0] == 0x03           # track_name meta first
    events = eng.analyze_array(audio_io.read_wav(wav, 22050))
    n_on = sum(1 for t in tracks for msg in t if msg[1] == 0x90)
    assert n_on == len(events) > 0
    y2, S_dB = eng.load_audio(wav)
    assert S_dB.shape[0] == 128 and len(eng.pitch_tracking(y2)[0]) == S_dB.shape[1]
    silent = str(tmp_path / "silent.wav")
    audio_io.write_wav(silent, np.zeros(22050, np.float32), 22050)
    assert eng.audio_to_midi_financial(silent, mid) is None


def test_batched_v2_engine_equals_clip_by_clip():
    """AegisFinancialEngine.analyze_arrays (one GPU batch, column means of the dB image from the GPU, fused pitch
    analysis, one RSI call) == analyze_array per clip, which test_financial_events_match_reference pins to the reference."""
    clips = [signals.guitar_clip(10.0, sr=22050, seed=21), signals.polyphonic_clip(6.0, sr=22050, seed=5),
             signals.guitar_test_track(sr=22050), signals.c_major_scale(22050), signals.guitar_clip(14.0, sr=22050, seed=3)]
    eng = AegisFinancialEngine()
    for kw in ({}, {"confidence_threshold": 0.6, "min_note_duration_ms": 80}, {"use_harmonic_filter": False, "sustain_ms": 120},
               {"use_guitar_filters": False}):
        batch = eng.analyze_arrays(clips, **kw)
        assert len(batch) == len(clips) and sum(len(b) for b in batch) > 10
        for y, ev in zip(clips, batch):
            one = eng.analyze_array(y, **kw)
            assert len(ev) == len(one)
            for a, b in zip(ev, one):
                assert set(a) == set(b)
                for k in b:
                    if k == "key_info":
                        assert (a[k]["key"], a[k]["mode"]) == (b[k]["key"], b[k]["mode"])
                    else:
                        assert a[k] == b[k] or (a[k] != a[k] and b[k] != b[k]), (k, a[k], b[k])
    # the column means of the dB image equal NumPy's on the image itself, bit for bit
    h = eng.handle
    res, bufs, off = h.analyze_batch(clips[:2], want_sdb=True, want_col_means=True, concatenated=True)
    F = int(off[-1])
    for i, r in enumerate(res):
        a, b = int(off[i]), int(off[i + 1])
        S = r["S_dB"]
        np.testing.assert_array_equal(bufs["sdb_col_means"][a:b], np.mean(S, axis=0))
        np.testing.assert_array_equal(bufs["sdb_col_means"][F + a:F + b], np.mean(S[:64], axis=0))
        np.testing.assert_array_equal(bufs["sdb_col_means"][2 * F + a:2 * F + b], np.mean(S[64:], axis=0))


def test_batched_v2_engine_with_clips_too_short_for_the_trend_filters():
    """A clip of fewer than 10 frames makes analyze_array raise IndexError, as the reference does (Bollinger window,
    financial_analysis.py:113-146); in a folder batch it must not take the other clips' results with it."""
    import pytest
    good = [signals.guitar_clip(8.0, sr=22050, seed=21), signals.c_major_scale(22050)]
    tiny, empty = good[0][:2000].copy(), np.zeros(0, np.float32)
    eng = AegisFinancialEngine()
    with pytest.raises(IndexError):
        eng.analyze_array(tiny)
    out = eng.analyze_arrays([good[0], tiny, good[1], empty], return_exceptions=True)
    assert isinstance(out[1], IndexError) and isinstance(out[3], IndexError)
    for got, y in zip((out[0], out[2]), good):
        one = eng.analyze_array(y)
        assert [(e["note"], e["start"], e["end"]) for e in got] == [(e["note"], e["start"], e["end"]) for e in one] and len(one) > 0
    with pytest.raises(IndexError) as ei:
        eng.analyze_arrays([good[0], tiny, good[1]])
    assert len(ei.value.results) == 3 and len(ei.value.results[2]) == len(out[2])
    assert eng.analyze_arrays([tiny], use_financial=False) == [eng.analyze_array(tiny, use_financial=False)]
