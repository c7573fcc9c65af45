"""Host-side mirror (midi_logic, SMF writer, Turbo chunking, WAV reader, sharding) against the
oracle's literal restatement of the reference.  CPU only: the raw_data comes from the oracle."""
import io
import os

import numpy as np
import pytest

from oracle import engine as oengine, smf as osmf
from spectrogram_midi_amd import audio_io, dist, midi_logic, smf
from tools import signals
from spectrogram_midi_amd.engine import AegisEngine


@pytest.fixture(scope="module")
def raws():
    return {"guitar": oengine.audio_to_midi(signals.guitar_test_track()),
            "notes": oengine.audio_to_midi(signals.guitar_clip(6.0, seed=11)),
            "scale": oengine.audio_to_midi(signals.c_major_scale(44100))}


KW = [{}, {"min_note_duration_ms": 100, "sustain_ms": 200}, {"noise_gate_db": -20, "confidence_threshold": 0.3},
      {"midi_program": 30, "vibrato_rate": 6.0, "vibrato_depth": 0.5, "turbo_mode": True, "start_time": 0}]


def same_events(a, b):
    """Every field identical; the articulation slope (closed-form least squares in the product, np.polyfit's SVD in the
    oracle and the reference) to 1e-9."""
    def same(k, u, v):
        if k == "slope":
            return abs(u - v) <= 1e-9 * max(1.0, abs(v))
        return u == v or u is v
    return len(a) == len(b) and all(set(x) == set(y) and all(same(k, x[k], y[k]) for k in x) for x, y in zip(a, b))


@pytest.mark.parametrize("kw", KW)
def test_events_and_midi_bytes_match_oracle(raws, kw):
    eng = AegisEngine()
    for name, raw in raws.items():
        ref_events, ref_blob = oengine.extract_events(raw, want_smf=True, **kw)
        buf = io.BytesIO()
        events = eng.extract_events(raw, buf, **kw)
        assert same_events(events, ref_events), name
        assert buf.getvalue() == ref_blob, name
        assert events and all(isinstance(e["note"], int) and isinstance(e["start"], int) for e in events)


def test_event_schema_and_file_output(raws, tmp_path):
    eng = AegisEngine()
    path = str(tmp_path / "out.mid")
    events = eng.extract_events(raws["notes"], path)
    assert set(events[0]) == {"note", "start", "end", "confidence", "velocity", "track", "rms_energy", "technique", "slope"}
    typ, tpb, tracks = osmf.parse_smf(open(path, "rb").read())
    assert (typ, tpb) == (1, 480)
    n_on = sum(1 for t in tracks for m in t if m[1] == 0x90)
    assert n_on == len(events)
    assert eng.extract_events(raws["notes"], None) == events            # no file requested


def test_truncation_to_shortest_array(raws):
    raw = dict(raws["guitar"])
    raw["f0"] = np.concatenate([raw["f0"], [440.0] * 8])                # Turbo-mode style length drift
    raw["voiced_flag"] = np.concatenate([raw["voiced_flag"], [True] * 8])
    raw["voiced_probs"] = np.concatenate([raw["voiced_probs"], [1.0] * 8])
    eng = AegisEngine()
    assert same_events(eng.extract_events(raw, None), oengine.extract_events(raws["guitar"]))


def test_empty_and_unvoiced_inputs():
    z = np.zeros(10)
    assert midi_logic.get_midi_events(z.astype(bool), z, z.astype(bool), z, np.ones(10, np.float32), 44100, 512, 0.7) == []
    assert midi_logic.get_midi_events(np.zeros(0, bool), np.zeros(0), np.zeros(0, bool), np.zeros(0),
                                      np.ones(1, np.float32), 44100, 512, 0.7) == []


def test_turbo_spans_follow_reference_rule():
    eng = AegisEngine()
    for cores in (1, 3, 8, 64):
        eng.turbo_cores = cores
        for n in (441000, 7938000, 300000, 5 * 44100 + 1):
            assert eng._turbo_spans(n) == oengine.turbo_chunks(n, 44100, 512, cores)


def test_wav_roundtrip(tmp_path):
    y = signals.sine_sweep(0.5)
    p = str(tmp_path / "a.wav")
    audio_io.write_wav(p, y, 44100)
    z = audio_io.read_wav(p, 44100)
    assert z.dtype == np.float32 and len(z) == len(y) and np.abs(z - y).max() <= 1.0 / 32768 + 1e-7
    part = audio_io.read_wav(p, 44100, offset=0.1, duration=0.2)
    np.testing.assert_array_equal(part, z[4410:4410 + 8820])
    with pytest.raises(ValueError):
        audio_io.read_wav(p, 22050, resample_mismatch=False)


def test_loader_resamples_like_librosa_polyphase(tmp_path):
    """librosa.load semantics: native-rate offset/duration, channel mean, then resample to
    ceil(n * ratio) samples.  The FIR is scipy's polyphase (librosa res_type="polyphase"), not soxr_hq."""
    import wave
    t = np.arange(22050) / 22050.0
    left, right = 0.5 * np.sin(2 * np.pi * 440 * t), 0.25 * np.sin(2 * np.pi * 880 * t)
    p = str(tmp_path / "stereo22k.wav")
    pcm = np.clip(np.round(np.stack([left, right], 1) * 32768.0), -32768, 32767).astype("<i2")
    with wave.open(p, "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(22050); w.writeframes(pcm.tobytes())
    with pytest.warns(UserWarning, match="resampling 22050 -> 44100"):
        y = audio_io.read_wav(p, 44100)
    assert y.dtype == np.float32 and len(y) == 44100
    t2 = np.arange(44100) / 44100.0
    want = 0.5 * (0.5 * np.sin(2 * np.pi * 440 * t2) + 0.25 * np.sin(2 * np.pi * 880 * t2))
    assert np.abs(y[2000:-2000] - want[2000:-2000]).max() < 1e-3
    with pytest.warns(UserWarning):
        part = audio_io.read_wav(p, 44100, offset=0.25, duration=0.5)
    assert len(part) == 22050
    assert len(audio_io.resample(np.zeros(1001, np.float32), 44100, 22050)) == 501      # ceil(n * ratio)
    np.testing.assert_array_equal(audio_io.read_wav_bytes(open(p, "rb").read(), 22050),
                                  audio_io.read_wav(p, 22050))


def test_cq_to_chroma_matches_oracle_and_structure():
    from oracle import chroma as ochroma
    from spectrogram_midi_amd import similarity
    m = similarity.cq_to_chroma(252)
    np.testing.assert_array_equal(m, ochroma.cq_to_chroma(252))
    assert m.shape == (12, 252) and m.dtype == np.float32
    assert (m.sum(axis=0) == 1).all() and (m.sum(axis=1) == 21).all()          # 7 octaves x 3 bins per class
    # C1 is bin 0: bins 251, 0, 1 (one third-semitone either side) fold onto pitch class C
    assert m[0, 0] == 1 and m[0, 1] == 1 and m[0, 35] == 1 and m[1, 2] == 1
    a = similarity.cq_to_chroma(84, bins_per_octave=12, fmin=440.0)             # A4 first: class 9 leads
    assert a[9, 0] == 1 and a[10, 1] == 1
    with pytest.raises(ValueError):
        similarity.cq_to_chroma(84, bins_per_octave=14)


def test_out_of_scope_methods_raise():
    eng = AegisEngine()
    for call in (lambda: eng.separate_stems("a.wav", "out"), lambda: eng.generate_tabs([]),
                 lambda: eng.export_musicxml({}, "x.xml")):
        with pytest.raises(NotImplementedError):
            call()
    assert AegisEngine.analyze is AegisEngine.audio_to_midi


def test_sharding_is_balanced_and_complete():
    rng = np.random.default_rng(0)
    dur = rng.uniform(30, 330, 512)
    shards = dist.shard_clips(dur, 8)
    assert sorted(i for s in shards for i in s) == list(range(512))
    loads = [dur[s].sum() for s in shards]
    assert max(loads) - min(loads) < 330
    assert dist.shard_clips([], 4) == [[], [], [], []]
    assert dist.shard_clips([5.0, 1.0], 1) == [[0, 1]]


def test_event_packing_roundtrip(raws):
    ev = oengine.extract_events(raws["notes"])
    back = dist.unpack_events(dist.pack_events(7, ev))[7]
    keys = ("note", "start", "end", "velocity", "track", "technique")
    assert [[e[k] for k in keys] for e in back] == [[e[k] for k in keys] for e in ev]
    assert np.allclose([e["confidence"] for e in back], [e["confidence"] for e in ev])


def test_randomised_frame_arrays_match_oracle():
    """Seeded fuzz of the vectorised event logic against the oracle's literal restatement: random voicing,
    pitch wobble, gaps, rake flags, levels and keyword settings (300 cases)."""
    rng = np.random.default_rng(20260220)
    eng = AegisEngine()
    for case in range(300):
        n = int(rng.integers(1, 400))
        midi = np.repeat(rng.integers(40, 84, n // 7 + 1), 7)[:n] + rng.normal(0, rng.choice([0.0, 0.05, 0.3]), n)
        f0 = 440.0 * 2 ** ((midi - 69) / 12)
        voiced = rng.random(n) < rng.choice([0.3, 0.8, 1.0])
        f0 = np.where(voiced, f0, 0.0)
        raw = {"f0": f0, "voiced_flag": voiced, "voiced_probs": rng.random(n),
               "rake_mask": rng.random(n) < 0.05, "rms": (rng.random(n) ** 3).astype(np.float32) + np.float32(1e-7)}
        kw = {"noise_gate_db": float(rng.choice([-40, -20, -60])), "sustain_ms": float(rng.choice([0, 50, 200])),
              "min_note_duration_ms": float(rng.choice([0, 50, 100])), "confidence_threshold": float(rng.choice([0.3, 0.7]))}
        ref_events, ref_blob = oengine.extract_events(raw, want_smf=True, **kw)
        buf = io.BytesIO()
        events = eng.extract_events(raw, buf, **kw)
        assert same_events(events, ref_events), case
        assert buf.getvalue() == ref_blob, case


def test_estimate_tuning_matches_oracle_and_known_answers():
    """librosa.estimate_tuning as cqt(tuning=None) uses it (auto_matcher.py:68-69): the product's host routine and the
    oracle's restatement agree exactly; a pure tone lands in the histogram cell of its deviation from the grid."""
    from oracle import chroma as ochroma
    from spectrogram_midi_amd import similarity
    t = np.arange(2 * 44100) / 44100
    for cents in (0, 7, -9, 14):
        y = (0.5 * np.sin(2 * np.pi * 440.0 * 2 ** (cents / 1200) * t)).astype(np.float32)
        tn = similarity.estimate_tuning(y, 44100, 36)
        assert tn == ochroma.estimate_tuning(y, 44100, bins_per_octave=36)
        want = cents / 100 * 3                              # bins of a third of a semitone
        err = (tn - want + 0.5) % 1.0 - 0.5                 # circular: the estimate lives in [-0.5, 0.5)
        # piptrack's parabolic interpolation of a Hann peak is a few cents off (librosa's behaviour, restated as is)
        assert abs(err) <= 0.16, (cents, tn, want)
    for y in (signals.guitar_clip(4.0, seed=5), signals.polyphonic_clip(4.0, seed=6), np.zeros(30000, np.float32)):
        assert similarity.estimate_tuning(y, 44100, 36) == ochroma.estimate_tuning(y, 44100, bins_per_octave=36)
