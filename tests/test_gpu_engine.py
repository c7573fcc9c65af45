"""End-to-end parity of the AegisEngine surface on the GPU against the oracle, plus the
size-independent properties used at BASELINE.json sizes.  Calls go through the C ABI."""
import io
import os

import numpy as np
import pytest

from oracle import engine as oengine, pyin as opyin, dsp as odsp, rake as orake
from spectrogram_midi_amd import _lib, audio_io
from tools import signals
from spectrogram_midi_amd.engine import AegisEngine
from spectrogram_midi_amd.worker import _pyin_worker

pytestmark = pytest.mark.gpu
EV_KEYS = ("note", "start", "end", "velocity", "track", "technique")


def assert_raw_equal(raw, ref, tag=""):
    np.testing.assert_array_equal(raw["voiced_flag"], ref["voiced_flag"], err_msg=tag)
    np.testing.assert_array_equal(raw["rake_mask"], ref["rake_mask"], err_msg=tag)
    np.testing.assert_array_equal(raw["rms"], ref["rms"], err_msg=tag)
    np.testing.assert_allclose(raw["f0"], ref["f0"], rtol=1e-13, err_msg=tag)
    np.testing.assert_allclose(raw["voiced_probs"], ref["voiced_probs"], rtol=1e-9, atol=1e-12, err_msg=tag)


def assert_events_equal(ev, ref, tag=""):
    assert [[e[k] for k in EV_KEYS] for e in ev] == [[e[k] for k in EV_KEYS] for e in ref], tag
    np.testing.assert_allclose([e["confidence"] for e in ev], [e["confidence"] for e in ref], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose([e["slope"] for e in ev], [e["slope"] for e in ref], rtol=1e-6, atol=1e-9)
    np.testing.assert_array_equal([e["rms_energy"] for e in ev], [e["rms_energy"] for e in ref])


@pytest.fixture(scope="module")
def eng():
    e = AegisEngine()
    yield e
    e.close()


def test_fixture_track_events_and_midi(eng, tmp_path):
    """BASELINE configs[0]: the reference's own synthetic guitar fixture through the file API."""
    y = signals.guitar_test_track()
    wav = str(tmp_path / "synthetic_guitar_test.wav")
    audio_io.write_wav(wav, y, 44100)
    y16 = audio_io.read_wav(wav, 44100)
    raw = eng.audio_to_midi(wav, None, rake_sensitivity=0.6)
    ref = oengine.audio_to_midi(y16)
    assert_raw_equal(raw, ref)
    np.testing.assert_array_equal(raw["y"], y16)
    buf = io.BytesIO()
    ev = eng.extract_events(raw, buf)
    ev_ref, blob_ref = oengine.extract_events(ref, want_smf=True)
    assert_events_equal(ev, ev_ref)
    assert buf.getvalue() == blob_ref
    yl, S_dB = eng.load_audio(wav)
    np.testing.assert_allclose(S_dB, ref["S_dB"], atol=2e-3)
    assert S_dB.shape == (128, 360) and S_dB.max() == 0.0 and S_dB.min() >= -80.0
    part = eng.audio_to_midi(wav, None, start_time=1.0, end_time=3.0)
    assert_raw_equal(part, oengine.audio_to_midi(audio_io.read_wav(wav, 44100, 1.0, 2.0)))


def test_sweep_and_empty(eng):
    y = signals.sine_sweep(10.0)
    assert_raw_equal(eng.analyze_array(y), oengine.audio_to_midi(y), "sweep")
    assert eng.analyze_array(np.zeros(0, np.float32)) is None
    out = eng.analyze_arrays([np.zeros(0, np.float32), y[:30000]])
    assert out[0] is None and len(out[1]["f0"]) == 59
    with pytest.raises(ValueError):
        eng.analyze_array(np.array([0.0, np.nan, 0.1], np.float32))


def test_hostile_signals_match_the_oracle(eng):
    """What a folder of real recordings holds and the synthetic guitar does not (tools/signals.py::hostile_clips: offsets,
    clipping, impulse trains, levels around the 1e-6 energy clamps of pitch.py, tones outside [fmin, fmax], beating, the
    Nyquist tone, denormals, a step, white noise): one ragged batch, every raw_data array and the events against the oracle."""
    clips = signals.hostile_clips()
    names = list(clips)
    raws, evs, _ = eng.audio_to_midi_batch([clips[k] for k in names])
    for k, raw, ev in zip(names, raws, evs):
        ref = oengine.audio_to_midi(clips[k])
        assert_raw_equal(raw, ref, k)
        assert_events_equal(ev, oengine.extract_events(ref), k)
    solo = eng.analyze_array(clips["clipped"])
    for key in ("f0", "voiced_flag", "voiced_probs", "rms", "rake_mask"):
        np.testing.assert_array_equal(solo[key], raws[names.index("clipped")][key], err_msg=key)


def test_audio_to_midi_batch_equals_the_per_clip_calls(eng, test_clips):
    """AegisEngine.audio_to_midi_batch: one GPU batch + one batched (C++) event extraction / SMF rendering == the
    reference's two calls per clip, dict for dict and byte for byte; non-finite audio is refused (on the device)."""
    clips = [test_clips[k] for k in ("guitar", "sweep", "empty", "tiny", "notes", "silence")]
    for kw in ({}, {"min_note_duration_ms": 100, "sustain_ms": 200, "midi_program": 30, "confidence_threshold": 0.3}):
        raws, evs, blobs = eng.audio_to_midi_batch(clips, **kw)
        assert len(raws) == len(evs) == len(blobs) == len(clips)
        for y, raw, ev, blob in zip(clips, raws, evs, blobs):
            if len(y) == 0:
                assert raw is None and ev == [] and blob is None
                continue
            ref = oengine.audio_to_midi(y)
            assert_raw_equal(raw, ref)
            ev_ref, blob_ref = oengine.extract_events(ref, want_smf=True, **kw)
            assert_events_equal(ev, ev_ref)
            assert blob == blob_ref
            buf = io.BytesIO()
            assert_events_equal(eng.extract_events(raw, buf, **kw), ev_ref)
            assert buf.getvalue() == blob_ref
    bad = test_clips["notes"].copy()
    bad[12345] = np.inf
    with pytest.raises(ValueError, match="not finite everywhere .clip 1, sample 12345"):
        eng.audio_to_midi_batch([test_clips["tiny"], bad])
    for poison, where in ((-np.inf, 777), (np.nan, 40001)):          # the kernels' clamps are NaN-safe: the call comes back with the verdict
        bad = test_clips["notes"].copy()
        bad[where] = poison
        with pytest.raises(ValueError, match=f"not finite everywhere .clip 0, sample {where}"):
            eng.audio_to_midi_batch([bad, test_clips["tiny"]])
    r, e, b = eng.audio_to_midi_batch([test_clips["notes"]], want_midi=False, turbo_mode=True)
    ref = oengine.audio_to_midi(test_clips["notes"], turbo_mode=True, num_cores=eng.turbo_cores or os.cpu_count())
    assert_events_equal(e[0], oengine.extract_events(ref)) and b[0] is None


def test_turbo_mode_matches_reference_chunking(eng):
    y = signals.guitar_clip(12.0, seed=3)
    for cores in (8, 3):
        eng.turbo_cores = cores
        f0, vf, vp = eng._parallel_pitch_tracking(y)
        rf0, rvf, rvp = oengine.parallel_pitch_tracking(y, num_cores=cores)
        assert len(f0) == len(rf0) > 1 + len(y) // 512            # one extra frame per chunk
        np.testing.assert_array_equal(vf, rvf)
        np.testing.assert_allclose(np.nan_to_num(f0), np.nan_to_num(rf0), rtol=1e-13)
    raw = eng.analyze_array(y, turbo_mode=True)
    ref = oengine.audio_to_midi(y, turbo_mode=True, num_cores=3)
    assert_raw_equal(raw, ref, "turbo")
    assert_events_equal(eng.extract_events(raw, None), oengine.extract_events(ref))
    short = y[: 4 * 44100]                                   # < 5 s: single-pass bypass (aegis_engine.py:189)
    a = eng._parallel_pitch_tracking(short)
    b = opyin.pyin(short)
    np.testing.assert_array_equal(a[1], b[1])
    eng.turbo_cores = None


def _sine(freq, seconds, sr=44100, amp=0.5):
    t = np.arange(int(sr * seconds)) / sr
    return (amp * np.sin(2 * np.pi * freq * t)).astype(np.float32)


@pytest.mark.parametrize("sr,hop", [(44100, 512), (22050, 512), (44100, 256)])
def test_pyin_initial_distribution_modes(sr, hop, test_clips):
    """aegis_config.pyin_init: librosa's unvoiced start (default) and the uniform start, both against the oracle in the
    same mode, on clips that are pitched from sample 0 -- the only frames the choice reaches.  44.1 kHz / 512 and
    22.05 kHz / 512 run the band kernels (H = 25, 50), hop 256 the generic kernel."""
    clips = {"sine880": _sine(880.0, 0.5, sr), "sine1000": _sine(1000.0, 0.5, sr), "sine110": _sine(110.0, 2.0, sr),
             "tiny": test_clips["tiny"], "pitched_start": signals.pitched_start_clip(sr),
             "guitar": signals.guitar_test_track(sr=sr), "silence": np.zeros(3000, np.float32)}
    got = {}
    for mode in ("unvoiced", "uniform"):
        h = _lib.Handle(sample_rate=sr, hop_length=hop, pyin_init=mode)
        assert h.param("pyin_init") == (1 if mode == "uniform" else 0)
        res = h.analyze_batch(list(clips.values()), stages=_lib.STAGE_PYIN)
        for (name, y), r in zip(clips.items(), res):
            f0, vf, vp = opyin.pyin(y, sr=sr, hop_length=hop, p_init=mode)
            np.testing.assert_array_equal(r["voiced_flag"], vf, err_msg=f"{mode} {name}")
            np.testing.assert_array_equal(np.nan_to_num(r["f0"]), np.nan_to_num(f0), err_msg=f"{mode} {name}")
            np.testing.assert_array_equal(r["voiced_prob"], vp, err_msg=f"{mode} {name}")
            got[mode, name] = r["voiced_flag"]
        # the streamed form of the same clip ends in the same arrays (the first column is built by the first push)
        st = h.open_stream(max_seconds=4.0)
        y = clips["pitched_start"]
        for a in range(0, len(y), 2048):
            st.push(y[a:a + 2048])
        out = st.close()
        st.free()
        np.testing.assert_array_equal(out["voiced_flag"], got[mode, "pitched_start"])
        h.close()
    # librosa's start never voices frame 0; the uniform start does where frame 0 carries a trough on the track
    for name in clips:
        assert not got["unvoiced", name][0], name
        np.testing.assert_array_equal(got["unvoiced", name][1:], got["uniform", name][1:], err_msg=name)
    if (sr, hop) == (44100, 512):
        assert got["uniform", "sine880"][0] and got["uniform", "sine1000"][0] and got["uniform", "pitched_start"][0]


def test_turbo_chunks_start_unvoiced_like_librosa():
    """Every Turbo chunk (aegis_engine.py:197-210) is its own pyin call: its first frame is where p_init acts."""
    y = _sine(880.0, 6.0)
    for mode in ("unvoiced", "uniform"):
        e = AegisEngine(pyin_init=mode)
        e.turbo_cores = 4
        f0, vf, vp = e._parallel_pitch_tracking(y)
        rf0, rvf, rvp = oengine.parallel_pitch_tracking(y, num_cores=4, p_init=mode)
        np.testing.assert_array_equal(vf, rvf)
        np.testing.assert_array_equal(np.nan_to_num(f0), np.nan_to_num(rf0))
        first = np.cumsum([0] + [1 + (b - a) // 512 for a, b in e._turbo_spans(len(y))])[:-1]
        assert vf[first].any() == (mode == "uniform")
        e.close()
    with pytest.raises(ValueError):
        _lib.Handle(pyin_init="voiced")


def test_dead_voiced_waves_skip_their_steps(eng):
    """viterbi.hip: a voiced wave with nothing but dead targets at an easy frame skips the step.  The rule is exact (every
    parity test in this directory runs with it); here: it fires, on tonal and on noisy material, and the counters add up."""
    h = eng.handle
    for y in (signals.guitar_clip(8.0, seed=2), signals.guitar_clip(6.0, seed=3, noise_dbfs=-12.0), np.zeros(44100, np.float32)):
        h.viterbi_stats(reset=True)
        r = h.analyze_batch([y], stages=_lib.STAGE_PYIN)[0]
        st = h.viterbi_stats(reset=True)
        F = len(r["f0"])
        assert st["wave_steps"] == 14 * (F - 1)
        assert 0 < st["skipped"] <= 7 * (F - 1) and st["list_only"] <= st["wave_steps"] - st["skipped"]
        f0, vf, vp = opyin.pyin(y)
        np.testing.assert_array_equal(r["voiced_flag"], vf)
        np.testing.assert_array_equal(np.nan_to_num(r["f0"]), np.nan_to_num(f0))
    assert st["skipped"] == 7 * (F - 1)          # silence: no voiced target is ever observed


@pytest.mark.parametrize("sr,hop", [(44100, 512), (22050, 256), (48000, 512)])
def test_cmnd_in_the_frame_kernel_equals_the_walk_in_pyin_obs(sr, hop, monkeypatch):
    """kernels.hip: the frame kernel's epilogue forms the CMND of a workgroup's 16 frames at once (what runs); with
    AEGIS_CMND_IN_FRAME=0 at create, pyin_obs_kernel walks the cumsum frame by frame as before (what the stage tests and lag
    ranges the epilogue cannot hold use).  Same operations in the same order: every output bit-identical, on a ragged batch
    whose workgroups straddle clips, with a clip shorter than one hop and an empty one."""
    rng = np.random.default_rng(5)
    clips = [signals.guitar_clip(2.0 + 0.37 * i, sr=sr, seed=20 + i) for i in range(5)]
    clips += [np.zeros(0, np.float32), (0.1 * rng.standard_normal(hop // 2)).astype(np.float32), signals.guitar_clip(1.0, sr=sr, seed=31, noise_dbfs=-10.0)]
    outs = []
    # what runs (CMND and trough lists from the frame kernel), the round-3 path (CMND from the frame kernel, troughs found by
    # pyin_obs_kernel: AEGIS_TROUGHS_IN_FRAME=0), and everything in pyin_obs_kernel
    for cmnd, troughs in (("1", "1"), ("1", "0"), ("0", "1")):
        monkeypatch.setenv("AEGIS_CMND_IN_FRAME", cmnd)
        monkeypatch.setenv("AEGIS_TROUGHS_IN_FRAME", troughs)
        h = _lib.Handle(sample_rate=sr, hop_length=hop)
        outs.append(h.analyze_batch(clips, stages=_lib.STAGE_PYIN))
        h.close()
    monkeypatch.delenv("AEGIS_TROUGHS_IN_FRAME")
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            for k in ("f0", "voiced_flag", "voiced_prob"):
                np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    f0, vf, vp = opyin.pyin(clips[1], sr=sr, hop_length=hop)
    np.testing.assert_array_equal(outs[0][1]["voiced_flag"], vf)
    np.testing.assert_array_equal(outs[0][1]["voiced_prob"], vp)


def test_pyin_worker(eng):
    chunk = signals.guitar_clip(3.0, seed=4)
    f0, vf, vp = _pyin_worker((chunk, 44100, 512))
    r = opyin.pyin(chunk)
    np.testing.assert_array_equal(vf, r[1])
    np.testing.assert_allclose(np.nan_to_num(f0), np.nan_to_num(r[0]), rtol=1e-13)
    assert np.isnan(f0[~vf]).all()


def test_detect_rake_patterns_against_reference_goldens(eng):
    """aegis_rake_patterns (GPU) vs masks produced by the reference's vision.py itself."""
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "rake_golden.npz"))
    for i, (n_mels, F, sr, hop, ratio) in enumerate(G["cases"]):
        h = _lib.Handle(sample_rate=int(sr), hop_length=int(hop))
        got = h.rake_patterns(G[f"S_{i}"], float(ratio))
        np.testing.assert_array_equal(got, G[f"mask_{i}"], err_msg=f"case {i}")
        h.close()
    np.testing.assert_array_equal(eng.detect_rake_patterns(G["S_0"]), orake.detect_rake_patterns(G["S_0"], 512, 44100, 0.6))


@pytest.mark.parametrize("sr,hop", [(22050, 512), (44100, 256), (48000, 512), (44100, 1024), (44100, 441)])
def test_other_rates_and_hops(sr, hop):
    """v2 engine rate (aegis_engine_financial.py:36) uses the H=50 band kernel; the others take
    the generic Viterbi kernel (transition widths 31, 51@48k, 101); hop 441 is not a multiple of 4 samples, so
    the YIN walk takes its scalar-load path."""
    y = signals.guitar_clip(4.0, sr=sr, seed=9)
    h = _lib.Handle(sample_rate=sr, hop_length=hop)
    r = h.analyze_batch([y])[0]
    f0, vf, vp = opyin.pyin(y, sr=sr, hop_length=hop)
    np.testing.assert_array_equal(r["voiced_flag"], vf)
    np.testing.assert_allclose(np.nan_to_num(r["f0"]), np.nan_to_num(f0), rtol=1e-13)
    np.testing.assert_array_equal(r["rms"], odsp.rms(y, hop_length=hop))
    S_dB = odsp.power_to_db(odsp.melspectrogram(y, sr=sr, hop_length=hop))
    np.testing.assert_allclose(r["S_dB"], S_dB, atol=2e-3)
    np.testing.assert_array_equal(r["rake_mask"], orake.detect_rake_patterns(S_dB, hop, sr, 0.6))
    h.close()


def test_three_minute_clip_midi_matches_oracle(eng):
    """BASELINE configs[1]: one 180 s 44.1 kHz clip, full path, MIDI diffed against the CPU oracle."""
    y = signals.guitar_clip(180.0, seed=1)
    raw = eng.analyze_array(y)
    ref = oengine.audio_to_midi(y)
    assert len(raw["f0"]) == 15504
    assert_raw_equal(raw, ref, "180s")
    ev, ev_ref = eng.extract_events(raw, None), oengine.extract_events(ref)
    assert len(ev) > 50
    assert_events_equal(ev, ev_ref, "180s")


def test_batch_properties_at_scale():
    """Size-independent properties on 64 clips of up to 30 s (configs[2] shape): results do not
    depend on batch composition, order, pass splitting or repetition."""
    rng = np.random.default_rng(0)
    clips = [signals.guitar_clip(float(rng.uniform(2.0, 30.0)), seed=100 + i) if i % 8 else
             signals.polyphonic_clip(float(rng.uniform(2.0, 30.0)), seed=100 + i) for i in range(16)]
    clips = clips * 4                                         # 64 clips, ragged
    h = _lib.Handle()
    a = h.analyze_batch(clips)
    b = h.analyze_batch(clips)
    perm = rng.permutation(len(clips))
    c = h.analyze_batch([clips[i] for i in perm])
    small = _lib.Handle(max_frames_per_pass=4000)             # forces many passes
    d = small.analyze_batch(clips)
    for i in range(len(clips)):
        for k in a[i]:
            np.testing.assert_array_equal(a[i][k], b[i][k], err_msg=f"rerun {i}/{k}")
            np.testing.assert_array_equal(a[i][k], d[i][k], err_msg=f"passes {i}/{k}")
            np.testing.assert_array_equal(a[i][k], a[i % 16][k], err_msg=f"duplicate {i}/{k}")
    for pos, i in enumerate(perm):
        for k in a[i]:
            np.testing.assert_array_equal(c[pos][k], a[i][k], err_msg=f"perm {i}/{k}")
    for i in (0, 5, 8):                                        # and spot-check against the oracle
        ref = oengine.audio_to_midi(clips[i])
        np.testing.assert_array_equal(a[i]["voiced_flag"], ref["voiced_flag"])
        np.testing.assert_array_equal(a[i]["rake_mask"], ref["rake_mask"])
        np.testing.assert_allclose(np.nan_to_num(a[i]["f0"]), ref["f0"], rtol=1e-13)
    with pytest.raises(_lib.AegisError):
        small.analyze_batch([np.zeros(4001 * 512, np.float32)])   # one clip larger than a pass
    h.close(); small.close()


def test_unaligned_clips_and_fewer_mel_bands():
    """Clips whose packed offsets are not multiples of 4 samples (scalar-load path of the YIN walk) and a
    64-band mel bank."""
    y1, y2 = signals.guitar_clip(1.0, seed=41)[:44101], signals.guitar_clip(1.5, seed=42)[:60003]
    h = _lib.Handle()
    a = h.analyze_batch([y1, y2])
    for r, y in zip(a, (y1, y2)):
        f0, vf, vp = opyin.pyin(y)
        np.testing.assert_array_equal(r["voiced_flag"], vf)
        np.testing.assert_array_equal(r["voiced_prob"], vp)
        np.testing.assert_array_equal(r["rms"], odsp.rms(y))
    h.close()
    h64 = _lib.Handle(n_mels=64)
    r = h64.analyze_batch([y2], stages=_lib.STAGE_MEL)[0]
    ref = odsp.power_to_db(odsp.melspectrogram(y2, n_mels=64))
    assert r["S_dB"].shape == ref.shape == (64, 118)
    np.testing.assert_allclose(r["S_dB"], ref, atol=2e-3)
    h64.close()


def test_time_chunk_pipeline_ragged(monkeypatch):
    """The two-stream time-chunk pipeline (frame stage of chunk k+1 under the Viterbi of chunk k, Viterbi column and
    pointer maps carried across launches, host samples copied chunk by chunk) must not change a single bit, whatever
    the chunk size and however ragged the batch: clips that end in different chunks, an empty clip, a clip shorter
    than the first chunk.  AEGIS_TIME_CHUNK=64 cuts the 12 s clip into ~17 launches; the default handle runs the
    same clips in one launch each."""
    clips = [signals.guitar_clip(12.0, seed=5), signals.guitar_clip(0.4, seed=6), np.zeros(0, np.float32),
             signals.polyphonic_clip(7.7, seed=7), signals.guitar_clip(3.0, seed=8)[:100003]]
    ref_h = _lib.Handle()
    ref = ref_h.analyze_batch(clips)
    # both cuts of a ragged pass: every clip into the same number of chunks, each proportional to its length (what runs),
    # and one time axis for all clips (AEGIS_PROPORTIONAL_CHUNKS=0: short clips end in early chunks)
    # ... and the two builds of the band Viterbi: AEGIS_DENSE=1 forces the register-capped one with four-wave observation
    # workgroups (what passes of >= 256 clips take) on this small batch
    for chunk, prop, dense in (("64", "1", "0"), ("64", "0", "0"), ("256", "1", "0"), ("256", "0", "0"), ("64", "1", "1"), ("256", "0", "1")):
        monkeypatch.setenv("AEGIS_TIME_CHUNK", chunk)
        monkeypatch.setenv("AEGIS_PROPORTIONAL_CHUNKS", prop)
        monkeypatch.setenv("AEGIS_DENSE", dense)
        h = _lib.Handle()
        got = h.analyze_batch(clips)
        again = h.analyze_batch(list(reversed(clips)))[::-1]
        for i in range(len(clips)):
            for k in ref[i]:
                np.testing.assert_array_equal(got[i][k], ref[i][k], err_msg=f"chunk {chunk}/{prop}/{dense} clip {i} {k}")
                np.testing.assert_array_equal(again[i][k], ref[i][k], err_msg=f"chunk {chunk}/{prop}/{dense} reversed clip {i} {k}")
        # the device-pointer entry: the one that takes the proportional cut (a pass fed from host memory keeps one time axis)
        dev = _analyze_on_device(h, clips)
        for k in ("f0", "voiced_flag", "voiced_prob", "rms", "rake_mask"):
            want = np.concatenate([np.asarray(ref[i][k]) for i in range(len(clips))])
            np.testing.assert_array_equal(dev[k].astype(want.dtype), want, err_msg=f"device entry, chunk {chunk}/{prop}/{dense} {k}")
        h.close()
    monkeypatch.delenv("AEGIS_TIME_CHUNK")
    monkeypatch.delenv("AEGIS_PROPORTIONAL_CHUNKS")
    monkeypatch.delenv("AEGIS_DENSE")
    o = oengine.audio_to_midi(clips[3])
    np.testing.assert_array_equal(ref[3]["voiced_flag"], o["voiced_flag"])
    np.testing.assert_allclose(np.nan_to_num(ref[3]["f0"]), o["f0"], rtol=1e-13)
    ref_h.close()


def test_full_size_batch_properties():
    """BASELINE.json configs[3] per-GPU shard at full size (64 clips x 180 s = 992 256 frames, the bench workload):
    size-independent properties instead of an oracle run -- a clip's arrays do not depend on what else is in the
    batch (three clips re-analysed alone, bit-identical), a second run reproduces the first, frame counts follow
    1 + N // hop, f0 is NaN exactly where the frame is unvoiced and sits on the pitch grid elsewhere, and the
    device-pointer entry used by bench.py agrees with the host-buffer entry."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import torch
    clips = bench.make_clips(64, 180.0, seed0=1)
    h = _lib.Handle()
    a = h.analyze_batch(clips, want_sdb=False)
    assert sum(len(r["f0"]) for r in a) == 992256 and all(len(r["f0"]) == 1 + len(c) // 512 for r, c in zip(a, clips))
    for i in (0, 31, 63):
        alone = h.analyze_batch([clips[i]], want_sdb=False)[0]
        for k in alone:
            np.testing.assert_array_equal(a[i][k], alone[k], err_msg=f"clip {i} {k}")
    grid = h.table("freqs")
    for i in (5, 40):
        r = a[i]
        assert np.array_equal(np.isnan(r["f0"]), ~r["voiced_flag"])
        assert np.isin(r["f0"][r["voiced_flag"]], grid).all()
        assert (r["voiced_prob"] >= 0).all() and (r["voiced_prob"] <= 1).all() and (r["rms"] >= 0).all()
    # device-resident entry (what bench.py times)
    dev = torch.device("cuda", 0)
    n = np.array([len(c) for c in clips], np.int64)
    off = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
    d_pcm = torch.from_numpy(np.concatenate(clips)).to(dev)
    F = 992256
    outs = {"f0": torch.empty(F, dtype=torch.float64, device=dev), "voiced_flag": torch.empty(F, dtype=torch.uint8, device=dev),
            "voiced_prob": torch.empty(F, dtype=torch.float64, device=dev), "rms": torch.empty(F, dtype=torch.float32, device=dev),
            "rake_mask": torch.empty(F, dtype=torch.uint8, device=dev)}
    for _ in range(2):                                       # second pass: idempotent
        h.analyze_batch_device(d_pcm.data_ptr(), off, {k: v.data_ptr() for k, v in outs.items()}, sync=True)
        np.testing.assert_array_equal(outs["voiced_flag"].cpu().numpy().astype(bool), np.concatenate([r["voiced_flag"] for r in a]))
        np.testing.assert_array_equal(outs["f0"].cpu().numpy(), np.concatenate([r["f0"] for r in a]))
        np.testing.assert_array_equal(outs["rms"].cpu().numpy(), np.concatenate([r["rms"] for r in a]))
        np.testing.assert_array_equal(outs["rake_mask"].cpu().numpy().astype(bool), np.concatenate([r["rake_mask"] for r in a]))
    h.close()


def test_two_frame_streams_large_batch(monkeypatch):
    """Passes of >= 128 clips alternate their time chunks' frame stage over two streams (chunk k+1's FFTs under chunk
    k's YIN / observation kernels).  With AEGIS_TIME_CHUNK=64 a batch of 130 short ragged clips runs that path; the
    default handle analyses the same clips without chunking.  Bit-identical."""
    rng = np.random.default_rng(3)
    base = signals.guitar_clip(8.0, seed=21)
    clips = [base[o:o + n].copy() for o, n in zip(rng.integers(0, 150000, 130), rng.integers(30000, 190000, 130))]
    clips[7] = np.zeros(0, np.float32)
    ref_h = _lib.Handle()
    ref = ref_h.analyze_batch(clips)
    monkeypatch.setenv("AEGIS_TIME_CHUNK", "64")
    h = _lib.Handle()
    got = h.analyze_batch(clips)
    monkeypatch.delenv("AEGIS_TIME_CHUNK")
    for i in range(len(clips)):
        for k in ref[i]:
            np.testing.assert_array_equal(got[i][k], ref[i][k], err_msg=f"clip {i} {k}")
    h.close(); ref_h.close()


def _analyze_on_device(h, clips):
    """The device-pointer entry (what bench.py times; the only entry that takes the CU-partitioned schedules)."""
    import torch
    dev = torch.device("cuda", 0)
    n = np.array([len(c) for c in clips], np.int64)
    off = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
    F = int(sum(1 + len(c) // 512 for c in clips))
    d_pcm = torch.from_numpy(np.concatenate(clips) if len(clips) else np.zeros(0, np.float32)).to(dev)
    outs = {"f0": torch.empty(F, dtype=torch.float64, device=dev), "voiced_flag": torch.empty(F, dtype=torch.uint8, device=dev),
            "voiced_prob": torch.empty(F, dtype=torch.float64, device=dev), "rms": torch.empty(F, dtype=torch.float32, device=dev),
            "rake_mask": torch.empty(F, dtype=torch.uint8, device=dev)}
    h.analyze_batch_device(d_pcm.data_ptr(), off, {k: v.data_ptr() for k, v in outs.items()}, sync=True)
    return {k: v.cpu().numpy() for k, v in outs.items()}


def _ragged_clips(seed, n):
    rng = np.random.default_rng(seed)
    base = signals.guitar_clip(8.0, seed=23)
    clips = [base[o:o + m].copy() for o, m in zip(rng.integers(0, 150000, n), rng.integers(30000, 190000, n))]
    clips[11] = np.zeros(0, np.float32)
    return clips


def test_balanced_schedule_small_chunks(monkeypatch):
    """Passes of 16..64 clips on the CU-partitioned streams run on equal chunks alternating over two frame streams, with
    ONE Viterbi launch that waits for a flag per chunk (aegis_api.hip "balanced passes", viterbi.hip next_run).  60 short
    ragged clips (one empty) through the device entry with AEGIS_BALANCED_CHUNK=64 (dozens of chunk hand-overs): the
    persistent launch, one launch per chunk (AEGIS_VITERBI_PERSISTENT=0) and the schedule switched off agree bit for bit,
    and with the host-buffer entry."""
    clips = _ragged_clips(5, 60)
    monkeypatch.setenv("AEGIS_BALANCED_CHUNK", "0")
    ref_h = _lib.Handle()
    ref = _analyze_on_device(ref_h, clips)
    host = ref_h.analyze_batch(clips)
    np.testing.assert_array_equal(ref["f0"], np.concatenate([r["f0"] for r in host]))
    np.testing.assert_array_equal(ref["rake_mask"].astype(bool), np.concatenate([r["rake_mask"] for r in host]))
    for env in ({"AEGIS_BALANCED_CHUNK": "64"}, {"AEGIS_BALANCED_CHUNK": "64", "AEGIS_VITERBI_PERSISTENT": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        h = _lib.Handle()
        h.set_profiling(True)
        got = _analyze_on_device(h, clips)
        for k in ref:
            np.testing.assert_array_equal(got[k], ref[k], err_msg=f"{env} {k}")
        launches = h.kernel_launches("viterbi")
        assert (launches == 1) if "AEGIS_VITERBI_PERSISTENT" not in env else (launches > 1)
        got = _analyze_on_device(h, clips)                 # the same handle again: chunk flags of a new generation
        for k in ref:
            np.testing.assert_array_equal(got[k], ref[k], err_msg=f"{env} second run {k}")
        h.close()
    ref_h.close()


def test_balanced_schedule_at_22050_hz(monkeypatch):
    """The same schedules at the v2 engine's rate (22 050 Hz: transition width 101, viterbi_band_kernel<50>): single
    Viterbi launch, one launch per chunk and the schedule switched off agree bit for bit."""
    clips = _ragged_clips(7, 24)
    monkeypatch.setenv("AEGIS_BALANCED_CHUNK", "0")
    ref_h = _lib.Handle(sample_rate=22050)
    ref = _analyze_on_device(ref_h, clips)
    for env in ({"AEGIS_BALANCED_CHUNK": "64"}, {"AEGIS_BALANCED_CHUNK": "64", "AEGIS_VITERBI_PERSISTENT": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        h = _lib.Handle(sample_rate=22050)
        h.set_profiling(True)
        got = _analyze_on_device(h, clips)
        for k in ref:
            np.testing.assert_array_equal(got[k], ref[k], err_msg=f"{env} {k}")
        assert int(h.debug_fetch("persistent_fallbacks")[0]) == 0
        launches = h.kernel_launches("viterbi")
        assert (launches == 1) if "AEGIS_VITERBI_PERSISTENT" not in env else (launches > 1)
        h.close()
    ref_h.close()


def test_persistent_viterbi_gives_up(monkeypatch):
    """The persistent launch's wait is bounded (0.1 s + 0.1 s per million frames of the pass): with one chunk's flag
    withheld (test hook) the kernel gives up instead of hanging, the handle falls back to one launch per chunk for the
    next 16 calls and repeats the call, and the results are the usual ones; then the single launch is tried again."""
    clips = _ragged_clips(6, 60)
    monkeypatch.setenv("AEGIS_BALANCED_CHUNK", "64")
    monkeypatch.setenv("AEGIS_TEST_DROP_CHUNK_SIGNAL", "3")
    h = _lib.Handle()
    assert int(h.debug_fetch("persistent_fallbacks")[0]) == 0
    a = _analyze_on_device(h, clips)
    assert int(h.debug_fetch("persistent_fallbacks")[0]) == 1
    monkeypatch.delenv("AEGIS_TEST_DROP_CHUNK_SIGNAL")
    monkeypatch.setenv("AEGIS_BALANCED_CHUNK", "0")
    ref_h = _lib.Handle()
    b = _analyze_on_device(ref_h, clips)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])
    a = _analyze_on_device(h, clips)                     # and again on the handle that fell back
    assert int(h.debug_fetch("persistent_fallbacks")[0]) == 1
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])
    import time
    for _ in range(14):                                  # still on the per-chunk schedule
        _analyze_on_device(h, clips)
    assert int(h.debug_fetch("persistent_fallbacks")[0]) == 1
    t0 = time.perf_counter()
    a = _analyze_on_device(h, clips)                     # the 16th call after the give-up tries the single launch again
    assert int(h.debug_fetch("persistent_fallbacks")[0]) == 2 and time.perf_counter() - t0 < 1.0      # (the hook still withholds the flag)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])
    h.close(); ref_h.close()


def _per_clip(dev, clips, hop=512):
    """Concatenated device-entry arrays -> per-clip dicts with the raw_data names and dtypes."""
    out, o = [], 0
    for c in clips:
        F = 1 + len(c) // hop
        out.append({"f0": dev["f0"][o:o + F], "voiced_flag": dev["voiced_flag"][o:o + F].astype(bool),
                    "voiced_probs": dev["voiced_prob"][o:o + F], "rms": dev["rms"][o:o + F],
                    "rake_mask": dev["rake_mask"][o:o + F].astype(bool)})
        o += F
    assert o == len(dev["f0"])
    return out


def _check_throughput_pass(monkeypatch, clips, solo, oracle, tag):
    """What the two >= 256-clip tests below share.  `clips` through the device-pointer entry on a default handle (pass
    size from the free device memory, auto-dense Viterbi build + four-wave observation workgroups, proportional chunks
    when ragged): ONE pass, the dense build, no fallback of the single launch; frame counts; f0 NaN exactly where
    unvoiced and on the pitch grid elsewhere; the clips `solo` bit-identical to the same clip analysed alone; the clips
    `oracle` equal to oracle.engine.audio_to_midi; and the whole batch bit-identical with AEGIS_DENSE=0 (128-register
    build, eight-wave observation workgroups) and with AEGIS_DENSE=0 AEGIS_PROPORTIONAL_CHUNKS=0 (one time axis)."""
    ragged = len({len(c) for c in clips}) > 1
    h = _lib.Handle()
    dev = _analyze_on_device(h, clips)
    assert h.param("last_passes") == 1 and h.param("last_dense") == 1, tag
    assert h.param("last_proportional") == (1 if ragged else 0) and h.param("last_chunks") > 2, tag
    assert int(h.debug_fetch("persistent_fallbacks")[0]) == 0
    grid = h.table("freqs")
    h.close()
    got = _per_clip(dev, clips)
    voiced = dev["voiced_flag"].astype(bool)
    assert np.array_equal(np.isnan(dev["f0"]), ~voiced), tag
    assert np.isin(dev["f0"][voiced], grid).all(), tag
    assert 0.05 < voiced.mean() < 0.98, tag
    assert (dev["voiced_prob"] >= 0).all() and (dev["voiced_prob"] <= 1).all() and (dev["rms"] >= 0).all(), tag
    for env in ({"AEGIS_DENSE": "0"}, {"AEGIS_DENSE": "0", "AEGIS_PROPORTIONAL_CHUNKS": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        h = _lib.Handle()
        other = _analyze_on_device(h, clips)
        assert h.param("last_passes") == 1 and h.param("last_dense") == 0 and int(h.debug_fetch("persistent_fallbacks")[0]) == 0
        assert h.param("last_proportional") == (1 if (ragged and len(env) == 1) else 0)
        h.close()
        for k in dev:
            np.testing.assert_array_equal(other[k], dev[k], err_msg=f"{tag}: {env} {k}")
        for k in env:
            monkeypatch.delenv(k)
    h = _lib.Handle()
    for i in solo:
        alone = h.analyze_batch([clips[i]], want_sdb=False)[0]
        for k, name in (("f0", "f0"), ("voiced_flag", "voiced_flag"), ("voiced_prob", "voiced_probs"), ("rms", "rms"), ("rake_mask", "rake_mask")):
            np.testing.assert_array_equal(got[i][name], alone[k], err_msg=f"{tag}: clip {i} alone, {k}")
    h.close()
    for i in oracle:
        ref = oengine.audio_to_midi(clips[i])
        g = dict(got[i], f0=np.nan_to_num(got[i]["f0"]))
        assert_raw_equal(g, ref, f"{tag}: clip {i} vs oracle")
    return dev


def test_folder_512_clips_one_dense_pass(monkeypatch):
    """BASELINE.json configs[3] at the size the driver's N = 1 bench line runs (bench.py --config folder: 512 clips,
    durations U(30, 330) s as folder_audio_collector.py:113 keeps them, 1/8 polyphonic, 1/8 noisy; 97 010 audio-s,
    8.36 M frames): one pass sized from the free device memory, the register-capped Viterbi build beside four-wave
    observation workgroups of eight frames per wave, proportional chunks -- the path no smaller test reaches."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    durations = bench.folder_durations(512)
    clips = bench.make_folder_clips(range(512), durations)
    assert sum(1 + len(c) // 512 for c in clips) == 8356000
    order = np.argsort(durations)
    poly = [i for i in order if i % 8 == 6]
    noisy = [i for i in order if i % 8 == 7]
    solo = sorted({int(order[0]), int(order[-1]), int(poly[len(poly) // 2]), int(noisy[len(noisy) // 2]), 0, 511})
    oracle = [int(order[0]), int(poly[0]), int(noisy[0])]
    assert all(durations[i] <= 60.0 for i in oracle)
    dev = _check_throughput_pass(monkeypatch, clips, solo, oracle, "folder")
    # the digest bench.py prints for its last timed step (`outputs_check`) is the digest of THESE outputs
    f_off = np.concatenate([[0], np.cumsum([1 + len(c) // 512 for c in clips])])
    assert bench.FOLDER_DIGEST_EXPECTED is not None and bench.outputs_digest(dev, f_off, range(512)) == bench.FOLDER_DIGEST_EXPECTED


def test_256_clips_of_180_s_one_dense_pass(monkeypatch):
    """The other >= 256-clip regime: 256 x 180 s (uniform lengths: no proportional cut, 3.97 M frames)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    clips = bench.make_clips(256, 180.0, seed0=1)
    _check_throughput_pass(monkeypatch, clips, [0, 9, 130, 255], [], "256 x 180 s")


def test_dense_pass_of_short_ragged_clips(monkeypatch):
    """256 clips of 5..60 s (one empty, one shorter than a hop) under default chunking: every chunk selects >= 4096 frames,
    so the observation kernel runs eight frames per wave on four-wave workgroups beside the register-capped Viterbi build
    (what a dense pass launches), on clips short enough for the oracle to check several of them."""
    rng = np.random.default_rng(11)
    base = [signals.guitar_clip(60.0, seed=51), signals.polyphonic_clip(60.0, seed=52), signals.guitar_clip(60.0, seed=53, noise_dbfs=-12.0)]
    clips = []
    for i in range(256):
        b = base[i % 3]
        n = int(rng.uniform(5.0, 60.0) * 44100)
        clips.append(np.ascontiguousarray(np.roll(b, -int(rng.integers(0, len(b))))[:n] * np.float32(rng.uniform(0.5, 1.0)), dtype=np.float32))
    clips[17] = np.zeros(0, np.float32)
    clips[18] = clips[18][:300]
    lens = np.array([len(c) for c in clips])
    short = [int(i) for i in np.argsort(lens)[2:5]]
    _check_throughput_pass(monkeypatch, clips, [17, 18, 0, 255, int(np.argmax(lens))], short + [18], "256 short ragged clips")


def test_a_process_holding_a_live_handle_exits_after_an_exception():
    """gpurun_out/call53.log (round 3): a script died with an AttributeError while a Handle was alive -- 22 050 Hz, profiling
    on, one host-buffer analyze_batch of a clip long enough for time chunks (the CU-masked streams exist) -- and then sat
    until `timeout` killed it: Handle.__del__ -> aegis_destroy -> hipStreamDestroy of the CU-masked Viterbi stream never
    returned.  The same script must now end with the exception's exit code within seconds, through
    the atexit hook (`raise`) and through an explicit `del` in a live interpreter (`del`)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for variant in ("raise", "del"):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "exit_hang_probe.py"), variant, "22050", "1", "80", "1"],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 1, (variant, r.returncode, r.stderr[-400:])
        assert "chunks" in r.stdout and "Error" in r.stderr and "end of script" not in r.stdout, (variant, r.stdout, r.stderr[-400:])


@pytest.mark.parametrize("sr", [44100, 22050])
def test_time_split_viterbi_equals_the_sequential_run(sr, monkeypatch):
    """viterbi.hip "Time-split Viterbi": clips cut into segments that run concurrently (speculative runs from a guessed
    column, lock-on runs from the previous segment's end column, stitched back-trace, verification of every decision the
    accumulated rounding bound could flip, sequential redo of what cannot be certified).  AEGIS_TIME_SPLIT=<steps> forces
    it with short segments; the outputs must be those of the sequential kernel (AEGIS_TIME_SPLIT=0) bit for bit: tonal,
    polyphonic and noisy clips, a clip with a long silence (no voiced note near its boundaries: lock-on late or never),
    pure silence (exact ties everywhere), a clip shorter than a segment, an empty one; both band kernels (H = 25, 50)."""
    clips = [signals.guitar_clip(40.0, sr=sr, seed=5), signals.polyphonic_clip(21.7, sr=sr, seed=7),
             signals.guitar_clip(15.0, sr=sr, seed=3, noise_dbfs=-12.0),
             np.concatenate([signals.guitar_clip(6.0, sr=sr, seed=5), np.zeros(int(sr * 14.0), np.float32), signals.guitar_clip(7.0, sr=sr, seed=6)]),
             np.zeros(int(sr * 9.0), np.float32), signals.guitar_clip(1.2, sr=sr, seed=8), np.zeros(0, np.float32),
             signals.sine_sweep(12.0, sr=sr)]
    monkeypatch.setenv("AEGIS_TIME_SPLIT", "0")
    ref_h = _lib.Handle(sample_rate=sr)
    ref = ref_h.analyze_batch(clips, stages=_lib.STAGE_PYIN)
    assert ref_h.param("split_passes") == 0
    ref_h.close()
    for seg, warm in (("256", "128"), ("512", "64"), ("1024", "128")):
        monkeypatch.setenv("AEGIS_TIME_SPLIT", seg)
        monkeypatch.setenv("AEGIS_SPLIT_WARMUP", warm)
        h = _lib.Handle(sample_rate=sr)
        got = h.analyze_batch(clips, stages=_lib.STAGE_PYIN)
        assert h.param("split_passes") == 1 and h.param("last_split_segments") > len(clips)
        print(f"sr {sr} segments of {seg} steps, warm-up {warm}: {h.param('last_split_segments')} segments, "
              f"{h.param('split_flagged_clips')} of {len(clips)} clips redone sequentially")
        for i in range(len(clips)):
            for k in ("f0", "voiced_flag", "voiced_prob"):
                np.testing.assert_array_equal(got[i][k], ref[i][k], err_msg=f"segments {seg}/{warm} clip {i} {k}")
        one = h.analyze_batch([clips[0]], stages=_lib.STAGE_PYIN)[0]            # a single clip, the other entry's schedule
        np.testing.assert_array_equal(one["f0"], ref[0]["f0"])
        dev = _analyze_on_device(h, clips) if sr == 44100 else None
        if dev is not None:
            np.testing.assert_array_equal(dev["f0"], np.concatenate([r["f0"] for r in ref]))
        assert h.param("split_flagged_clips") <= 3 * 3                           # the silent clips at most (every call)
        h.close()
    monkeypatch.delenv("AEGIS_TIME_SPLIT")
    monkeypatch.delenv("AEGIS_SPLIT_WARMUP")
    # automatic: a pass bound by the recurrence of its longest clip is split (one long clip; a few), a pass of many short
    # clips is not
    h = _lib.Handle(sample_rate=sr)
    long_clip = signals.guitar_clip(100.0, sr=sr, seed=12)
    monkeypatch.setenv("AEGIS_TIME_SPLIT", "0")
    seq_h = _lib.Handle(sample_rate=sr)
    want = seq_h.analyze_batch([long_clip, clips[1]], stages=_lib.STAGE_PYIN)
    seq_h.close()
    monkeypatch.delenv("AEGIS_TIME_SPLIT")
    got = h.analyze_batch([long_clip, clips[1]], stages=_lib.STAGE_PYIN)
    assert h.param("split_passes") == 1 and h.param("last_split_segments") >= 4
    for a, b in zip(got, want):
        for k in ("f0", "voiced_flag", "voiced_prob"):
            np.testing.assert_array_equal(a[k], b[k], err_msg=f"automatic split {k}")
    h.analyze_batch([clips[5]] * 40, stages=_lib.STAGE_PYIN)
    assert h.param("split_passes") == 1 and h.param("last_split_segments") == 0
    h.close()


def test_time_split_rounds_of_second_speculation(monkeypatch):
    """Segments shorter than the recurrence takes to converge (512 steps, warm-up 128, a 180 s clip): some lock-on runs reach
    the end of their segment without meeting the speculative run, the segments behind them speculate again from the exact
    column that run left (viterbi_band.inc phases 3 / 4) -- and the outputs are still the sequential pass's, bit for bit,
    without a clip handed back to the sequential kernel."""
    y = signals.guitar_clip(180.0, seed=3)
    monkeypatch.setenv("AEGIS_TIME_SPLIT", "0")
    h = _lib.Handle()
    ref = _analyze_on_device(h, [y])
    h.close()
    rounds = []
    for seglen, warmup in ((512, 128), (256, 256), (384, 256)):
        monkeypatch.setenv("AEGIS_TIME_SPLIT", str(seglen))
        monkeypatch.setenv("AEGIS_SPLIT_WARMUP", str(warmup))
        h = _lib.Handle()
        got = _analyze_on_device(h, [y])
        assert h.param("last_split_segments") >= 30 and h.param("split_flagged_clips") == 0
        rounds.append(h.param("split_rounds"))
        lock = h.debug_fetch("seg_lock")
        assert (lock != -1).all()
        h.close()
        for k in ref:
            np.testing.assert_array_equal(got[k], ref[k], err_msg=f"{k} at {seglen}/{warmup}")
    assert max(rounds) >= 1, rounds


def test_time_split_on_one_rank_of_the_folder(monkeypatch):
    """Rank 0's shard of BASELINE.json configs[3] on 8 GPUs (64 ragged clips of 81 .. 328 s: tonal, polyphonic and noisy ones;
    1.04 M frames) with the time split forced on every clip: 250-odd concurrent segments, thousands of tubes resolved by the
    exact walk, the noisy clips (unvoiced throughout: the two edge bins' unvoiced states tie for the whole clip) resolved
    through rails and rounds of second speculation instead of being handed back to the sequential kernel -- and every output array equal
    to the sequential pass's, bit for bit."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from spectrogram_midi_amd import dist as adist
    durations = bench.folder_durations(512)
    mine = adist.shard_clips(durations, 8)[0]
    clips = bench.make_folder_clips(mine, durations)
    monkeypatch.setenv("AEGIS_TIME_SPLIT", "0")
    h = _lib.Handle()
    ref = _analyze_on_device(h, clips)
    h.close()
    monkeypatch.setenv("AEGIS_TIME_SPLIT", "4096")
    h = _lib.Handle()
    got = _analyze_on_device(h, clips)
    assert h.param("split_passes") == 1 and h.param("last_split_segments") > 200
    flagged = h.param("split_flagged_clips")
    v = h.debug_fetch("split_verify")
    assert flagged <= 2 and v[12] > 2000 and v[13] > 500                 # tubes resolved by the exact walk; the path changed in many
    assert v[14] >= 1 and v[15] > 10000                                  # rails: tubes as long as an unvoiced clip, scanned in parallel
    lock = h.debug_fetch("seg_lock")
    assert (lock != -1).all()                                            # a lock-on run that never met its speculative run was followed by a round of second speculation
    h.close()
    monkeypatch.delenv("AEGIS_TIME_SPLIT")
    for k in ref:
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)


@pytest.mark.parametrize("sr", [44100, 22050])
def test_time_split_pass_fed_from_host_memory_chunks_its_frame_stage(sr, monkeypatch):
    """The host-buffer entry (aegis_analyze_batch) on a pass the planner splits in time: the samples are copied and the frame
    stage runs chunk by chunk (each copy under the frame stage of the chunk before), the segments are launched once behind
    the last chunk -- same arrays as the sequential pass and as the device-resident split pass, bit for bit."""
    clips = [signals.guitar_clip(40.0 + 7 * i, sr=sr, seed=60 + i) if i % 3 else signals.polyphonic_clip(40.0 + 7 * i, sr=sr, seed=60 + i)
             for i in range(6)]
    monkeypatch.setenv("AEGIS_TIME_SPLIT", "0")
    h = _lib.Handle(sample_rate=sr)
    ref = h.analyze_batch(clips, want_sdb=False)
    assert h.param("last_split_segments") == 0
    h.close()
    monkeypatch.setenv("AEGIS_TIME_SPLIT", "512")
    monkeypatch.setenv("AEGIS_FEED_CHUNK", "96")               # x 64 / 6 clips = 1 024-step chunks
    h = _lib.Handle(sample_rate=sr)
    got = h.analyze_batch(clips, want_sdb=False)
    assert h.param("last_split_segments") >= 12 and h.param("last_chunks") >= 3 and h.param("last_passes") == 1
    assert h.param("split_flagged_clips") == 0
    again = h.analyze_batch(clips[::-1], want_sdb=False)[::-1]
    h.close()
    # ... and in the hybrid form (the sequential kernel chunk by chunk behind the copies up to step S, segments behind it)
    monkeypatch.setenv("AEGIS_SPLIT_HYBRID", "1")
    monkeypatch.setenv("AEGIS_HYBRID_PCT", "200")
    monkeypatch.setenv("AEGIS_FEED_CHUNK", "48")               # x 64 / 6 clips = 512-step chunks
    h = _lib.Handle(sample_rate=sr)
    hyb = h.analyze_batch(clips + clips[:3], want_sdb=False)   # nine clips
    S = h.param("last_hybrid_step")
    assert h.param("last_split_segments") >= 12 and h.param("split_flagged_clips") == 0
    h.close()
    for r, g, a in zip(ref, got, again):
        for k in r:
            np.testing.assert_array_equal(g[k], r[k], err_msg=k)
            np.testing.assert_array_equal(a[k], r[k], err_msg=k)
    for r, g in zip(ref + ref[:3], hyb):
        for k in r:
            np.testing.assert_array_equal(g[k], r[k], err_msg=f"hybrid {k} S={S}")


@pytest.mark.parametrize("persistent", ["1", "0"])
def test_hybrid_split_pass_equals_the_sequential_run(persistent, monkeypatch):
    """A split pass of <= 64 clips as the planner runs it on a rank's shard of the folder (aegis_api.hip split_hybrid): the
    balanced pipeline -- frame stage on 192 CUs, the sequential kernel on 64 -- until the frame stage is through, and only
    the steps behind the step S the sequential kernel has reached by then cut into speculative segments.  48 ragged clips
    (tonal, polyphonic, noisy, one shorter than S, one silent): every output array equal to the sequential pass's."""
    rng = np.random.default_rng(21)
    base = [signals.guitar_clip(150.0, seed=71), signals.polyphonic_clip(150.0, seed=72), signals.guitar_clip(150.0, seed=73, noise_dbfs=-12.0)]
    clips = []
    for i in range(48):
        n = int(rng.uniform(20.0, 150.0) * 44100)
        a = int(rng.integers(0, len(base[0]) - n + 1))
        clips.append(np.ascontiguousarray(base[i % 3][a:a + n]))
    clips[5] = clips[5][:44100 * 8]                          # ends long before S
    clips[6] = np.zeros(44100 * 100, np.float32)             # unvoiced throughout: rails
    monkeypatch.setenv("AEGIS_TIME_SPLIT", "0")
    h = _lib.Handle()
    ref = _analyze_on_device(h, clips)
    h.close()
    monkeypatch.setenv("AEGIS_TIME_SPLIT", "640")
    monkeypatch.setenv("AEGIS_SPLIT_HYBRID", "1")
    monkeypatch.setenv("AEGIS_VITERBI_PERSISTENT", persistent)
    h = _lib.Handle()
    got = _analyze_on_device(h, clips)
    S = h.param("last_hybrid_step")
    assert 2048 <= S < max(len(c) for c in clips) // 512 and h.param("last_split_segments") > 48 + 20
    assert h.param("last_persistent") == int(persistent) and int(h.debug_fetch("persistent_fallbacks")[0]) == 0
    assert h.param("split_flagged_clips") <= 1
    assert sum(1 for c in clips if len(c) // 512 <= S) >= 3              # clips the sequential kernel finished on its own
    again = _analyze_on_device(h, clips)
    h.close()
    for k in ref:
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
        np.testing.assert_array_equal(again[k], ref[k], err_msg=k)


def test_hybrid_split_pass_of_more_than_64_clips(monkeypatch):
    """The hybrid form on the UN-partitioned streams (a pass of 65 .. 255 clips: what a rank of four holds): the ramp's chunks
    with a launch of the sequential kernel per chunk up to step S, the speculative runs on the frame stream behind its last
    kernel.  72 ragged clips; forced segments, and the planner's own second rule on a batch it applies to."""
    rng = np.random.default_rng(23)
    base = [signals.guitar_clip(200.0, seed=91), signals.polyphonic_clip(200.0, seed=92), signals.guitar_clip(200.0, seed=93, noise_dbfs=-12.0)]
    clips = []
    for i in range(72):
        n = int(rng.uniform(20.0, 200.0) * 44100)
        a = int(rng.integers(0, len(base[0]) - n + 1))
        clips.append(np.ascontiguousarray(base[i % 3][a:a + n]))
    monkeypatch.setenv("AEGIS_TIME_SPLIT", "0")
    h = _lib.Handle()
    ref = _analyze_on_device(h, clips)
    h.close()
    for env in ({"AEGIS_TIME_SPLIT": "640", "AEGIS_SPLIT_HYBRID": "1"}, {}):
        monkeypatch.delenv("AEGIS_TIME_SPLIT", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        h = _lib.Handle()
        got = _analyze_on_device(h, clips)
        S = h.param("last_hybrid_step")
        if env:
            assert S >= 2048 and h.param("last_split_segments") > 72 + 50 and h.param("last_balanced") == 0 and h.param("last_persistent") == 0
        assert h.param("split_flagged_clips") <= 1
        auto_segments = h.param("last_split_segments")
        h.close()
        for k in ref:
            np.testing.assert_array_equal(got[k], ref[k], err_msg=f"{k} {env}")
        for k in env:
            monkeypatch.delenv(k)
    assert auto_segments >= 0          # (whether the second rule takes this batch depends on its longest clip; either way the outputs are the sequential pass's)


def test_hybrid_split_pass_at_22050_hz(monkeypatch):
    """The same at the v2 engine's rate (H = 50 band kernel, 7.3 us per step): a device-resident batch of 64 clips is where the
    planner takes a hybrid pass there.  Device entry of a 22 050 Hz handle, 64 ragged clips of 60 .. 180 s."""
    import torch
    sr = 22050
    rng = np.random.default_rng(22)
    base = [signals.guitar_clip(180.0, sr=sr, seed=81), signals.polyphonic_clip(180.0, sr=sr, seed=82), signals.guitar_clip(180.0, sr=sr, seed=83, noise_dbfs=-12.0)]
    clips = []
    for i in range(64):
        n = int(rng.uniform(60.0, 180.0) * sr)
        a = int(rng.integers(0, len(base[0]) - n + 1))
        clips.append(np.ascontiguousarray(base[i % 3][a:a + n]))
    dev = torch.device("cuda", 0)
    n = np.array([len(c) for c in clips], np.int64)
    off = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
    F = int(sum(1 + len(c) // 512 for c in clips))
    d_pcm = torch.from_numpy(np.concatenate(clips)).to(dev)

    def run(h):
        outs = {"f0": torch.empty(F, dtype=torch.float64, device=dev), "voiced_flag": torch.empty(F, dtype=torch.uint8, device=dev),
                "voiced_prob": torch.empty(F, dtype=torch.float64, device=dev), "rms": torch.empty(F, dtype=torch.float32, device=dev),
                "rake_mask": torch.empty(F, dtype=torch.uint8, device=dev)}
        h.analyze_batch_device(d_pcm.data_ptr(), off, {k: v.data_ptr() for k, v in outs.items()}, sync=True)
        return {k: v.cpu().numpy() for k, v in outs.items()}

    monkeypatch.setenv("AEGIS_TIME_SPLIT", "0")
    h = _lib.Handle(sample_rate=sr)
    ref = run(h)
    h.close()
    monkeypatch.delenv("AEGIS_TIME_SPLIT")
    monkeypatch.setenv("AEGIS_HYBRID_PCT", "130")
    h = _lib.Handle(sample_rate=sr)
    got = run(h)                                             # the automatic rule
    S = h.param("last_hybrid_step")
    assert S >= 2048 and h.param("last_split_segments") > 64 and h.param("split_flagged_clips") <= 1
    assert int(h.debug_fetch("persistent_fallbacks")[0]) == 0
    h.close()
    for k in ref:
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)


def test_out_of_memory_retry_halves_the_passes():
    """An analyze call whose workspace cannot be allocated (another handle or the caller took the memory the pass size was
    derived from) halves max_frames_per_pass -- down to 2^21 frames -- and plans its passes again instead of failing: the first
    workspace growth of a fresh handle fails through the test hook, the call succeeds with the same results, the pass size is
    halved; at the floor the error comes out as AEGIS_ERR_NOMEM and the handle stays usable."""
    clips = [signals.guitar_clip(6.0 + i, seed=40 + i) for i in range(3)]
    ref_h = _lib.Handle()
    want = _analyze_on_device(ref_h, clips)
    ref_h.close()
    h = _lib.Handle()
    before = h.param("max_frames_per_pass")
    assert before > 2 ** 21
    h.lib.aegis_debug_fetch(h._h, b"fail_allocs", None, 1)
    got = _analyze_on_device(h, clips)
    assert h.param("max_frames_per_pass") == max(2 ** 21, before // 2)
    for k in want:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    h.close()
    h = _lib.Handle(max_frames_per_pass=2 ** 21)
    h.lib.aegis_debug_fetch(h._h, b"fail_allocs", None, 1)
    with pytest.raises(_lib.AegisError) as ei:
        _analyze_on_device(h, clips)
    assert ei.value.code == _lib.ERR_NOMEM
    got = _analyze_on_device(h, clips)
    np.testing.assert_array_equal(got["f0"], want["f0"])
    h.close()


def test_graft_entry_smoke():
    """The driver's smoke() hook itself: one small analyze on cuda:0 checked against the oracle."""
    import __graft_entry__ as g
    g.smoke()
