"""Pins the CPU oracle (oracle/) with analytic known answers -- the anchors SURVEY.md 8c lists in
lieu of librosa golden vectors (the reference holds none; librosa is absent).  CPU only."""
import numpy as np
import pytest

from oracle import dsp, engine as oengine, pyin as opyin

SR = 44100


def sine(freq, seconds, amp=0.5):
    t = np.arange(int(SR * seconds)) / SR
    return (amp * np.sin(2 * np.pi * freq * t)).astype(np.float32)


def test_note_frequencies():
    assert dsp.note_to_hz("E2") == 82.4068892282175
    assert dsp.note_to_hz("C6") == 1046.5022612023945
    assert dsp.hz_to_midi(440.0) == 69.0


def test_pyin_geometry():
    p = opyin.PyinParams()
    assert (p.min_period, p.max_period, p.n_lags) == (42, 536, 495)
    assert p.n_pitch_bins == 441 and p.transition_width == 51 and p.max_semitones_per_frame == 5
    assert 12 * 10 * np.log2(p.fmax / p.fmin) == 440.0
    p2 = opyin.PyinParams(sr=22050)
    assert (p2.min_period, p2.max_period, p2.n_lags, p2.transition_width) == (21, 268, 248, 101)
    T = opyin.transition_matrix(p)
    assert T.shape == (882, 882) and np.allclose(T.sum(axis=1), 1.0)
    assert np.count_nonzero(T[300]) == 102           # 51-wide band x 2 voicing blocks


def test_frame_count_and_padding():
    for n in (0, 1, 511, 512, 513, 44100):
        y = np.zeros(n, np.float32)
        assert len(dsp.rms(y)) == 1 + n // 512
        assert dsp.melspectrogram(y).shape == (128, 1 + n // 512)


def test_pure_sine_tracks_a2():
    """Every frame is pinned, the first ones included: frame 0 (half zero padding, voiced_prob 0.01) starts the chain
    unvoiced at the A2 bin's twin, frame 1 (three quarters of a window) lands one bin high, everything else sits on
    bin 50 = 110 Hz until the zero-padded last frame (one bin high again)."""
    p = opyin.PyinParams()
    for mode in ("unvoiced", "uniform"):
        f0, voiced, prob, it = opyin.pyin(sine(110.0, 2.0), p_init=mode, return_intermediates=True)
        states = it["states"].astype(int)
        assert len(states) == 173
        assert states[0] == 441 + 51 and not voiced[0] and np.isnan(f0[0])
        assert states[1] == 51 and states[-1] == 51 and np.all(states[2:-1] == 50)
        assert voiced[1:].all() and np.all(f0[2:-1] == p.freqs[50])      # fmin * 2^(50/120) = 110 Hz bin
        assert abs(p.freqs[50] - 110.0) < 1e-9
        assert prob[0] == 0.01 and np.all(prob[2:-1] > 0.99)


def test_initial_distribution_is_librosas_unvoiced_start():
    """core/pitch.py::pyin: p_init = zeros(2B); p_init[B:] = 1/B.  With it no clip and no Turbo chunk can start voiced
    (a voiced first state costs log(tiny)); the uniform alternative voices frame 0 of a clip that is pitched from
    its first sample."""
    p = opyin.PyinParams()
    a = opyin.initial_distribution(p)
    assert a.shape == (882,) and np.all(a[:441] == 0) and np.all(a[441:] == 1 / 441) and abs(a.sum() - 1) < 1e-12
    assert np.all(opyin.initial_distribution(p, "uniform") == 1 / 882)
    with pytest.raises(ValueError):
        opyin.initial_distribution(p, "voiced")
    for f in (880.0, 1000.0):
        y = sine(f, 0.5)
        lib = opyin.pyin(y)                           # default = librosa's
        uni = opyin.pyin(y, p_init="uniform")
        assert not lib[1][0] and uni[1][0]            # frame 0: unvoiced vs voiced
        assert np.array_equal(lib[1][1:], uni[1][1:]) and np.array_equal(lib[2], uni[2])
        assert np.array_equal(np.nan_to_num(lib[0][1:]), np.nan_to_num(uni[0][1:]))
    # direct: observations that favour a voiced bin at t = 0 (0.6 against (1 - 0.6)/441 per unvoiced state)
    obs = np.zeros((882, 3))
    obs[100, :] = 0.6
    obs[441:, :] = 0.4 / 441
    assert list(opyin.decode(obs, p, p_init="uniform")) == [100, 100, 100]
    assert list(opyin.decode(obs, p)) == [441 + 100, 100, 100]
    assert list(opyin.decode(obs, p, use_c=False)) == [441 + 100, 100, 100]


def test_turbo_chunks_start_unvoiced():
    """aegis_engine.py:197-210: every Turbo chunk is its own pyin call, so every chunk of a sustained note starts
    with an unvoiced frame under librosa's start."""
    y = sine(880.0, 6.0)
    f0, voiced, prob = oengine.parallel_pitch_tracking(y, num_cores=4)
    spans = oengine.turbo_chunks(len(y), SR, 512, 4)
    first, k = [], 0
    for lo, hi in spans:
        first.append(k)
        k += 1 + (hi - lo) // 512
    assert k == len(voiced) and not voiced[first].any()
    uni = oengine.parallel_pitch_tracking(y, num_cores=4, p_init="uniform")
    assert uni[1][first].any()                         # the uniform start voices the chunk that begins on a zero crossing
    rest = np.setdiff1d(np.arange(k), first)
    assert np.array_equal(voiced[rest], uni[1][rest])


def test_silence_is_unvoiced():
    f0, voiced, prob = opyin.pyin(np.zeros(SR, np.float32))
    assert not voiced.any() and np.isnan(f0).all() and np.all(prob == 0)


def test_rms_of_sine():
    r = dsp.rms(sine(441.0, 1.0, amp=0.5))            # 441 Hz: 100-sample period
    assert np.allclose(r[4:-4], 0.5 / np.sqrt(2), rtol=2e-3)


def test_mel_filterbank_slaney_area():
    fb = dsp.mel_filterbank(SR, 2048)
    assert fb.shape == (128, 1025) and fb.dtype == np.float32 and (fb >= 0).all()
    df = SR / 2048
    area = fb.sum(axis=1) * df                         # slaney norm: unit area per band
    assert np.allclose(area[5:], 1.0, rtol=0.15)
    assert np.all(np.diff(np.argmax(fb, axis=1)) >= 0)


def test_stft_parseval():
    rng = np.random.default_rng(0)
    y = rng.normal(0, 0.1, 8192).astype(np.float32)
    D = dsp.stft(y)
    win = dsp.hann_periodic(2048)
    frames = dsp.frame_centered(y, 2048, 512).astype(np.float64) * win[:, None]
    full = np.abs(D.astype(np.complex128)) ** 2
    energy = (full[0] + full[-1] + 2 * full[1:-1].sum(axis=0)) / 2048
    assert np.allclose(energy, (frames ** 2).sum(axis=0), rtol=1e-5)


def test_db_range():
    S = dsp.melspectrogram(sine(220.0, 1.0))
    db = dsp.power_to_db(S)
    assert db.max() == 0.0 and db.min() >= -80.0 and db.dtype == np.float32
    a = dsp.amplitude_to_db(np.array([1e-9, 0.5, 1.0], np.float32))
    assert a[2] == 0.0 and a[0] == -80.0 and abs(a[1] + 6.0206) < 1e-3


def test_c_and_numpy_viterbi_agree():
    y = np.concatenate([sine(110, 0.6), np.zeros(4000, np.float32), sine(330, 0.5, 0.2)])
    a = opyin.pyin(y, use_c=True)
    b = opyin.pyin(y, use_c=False)
    assert np.array_equal(a[1], b[1]) and np.array_equal(np.nan_to_num(a[0]), np.nan_to_num(b[0]))


def test_turbo_chunking_rule():
    # aegis_engine.py:192-204 with 8 cores on a 10 s clip: 862 ceil-frames -> 107 per core
    spans = oengine.turbo_chunks(441000, SR, 512, 8)
    assert len(spans) == 8 and spans[0] == (0, 107 * 512) and spans[-1] == (7 * 107 * 512, 441000)
    assert oengine.turbo_chunks(100, SR, 512, 8) == [(0, 100)]


def test_tick_rule_and_smf_header():
    from oracle import smf
    ev = [{"note": 45, "start": 10, "end": 40, "confidence": 0.9, "velocity": 100, "track": "main",
           "rms_energy": -3.0, "technique": None, "slope": 0.0}]
    blob = smf.write_smf(ev, SR, 512)
    typ, tpb, tracks = smf.parse_smf(blob)
    assert (typ, tpb, len(tracks)) == (1, 480, 2)
    on_tick = int(10 * (512 / SR) * 960)
    assert tracks[0][0][1] == 0xC0 and tracks[0][1] == (on_tick, 0x90, bytes([45, 100]))
    assert tracks[0][-1][1] == 0xFF and tracks[1][-1][1] == 0xFF      # end_of_track appended


def test_recursive_cqt_stays_close_to_the_direct_transform():
    """librosa computes the constant-Q transform octave by octave on resampled copies with sparsified FFT-domain filters
    (oracle/cqt_recursive.py restates it); the GPU computes the direct transform (oracle/cqt.py).  The two agree to under
    1 % of the peak magnitude and the chroma to 0.02: the bound DESIGN.md section 3.8 quotes (profiles/r3_cqt_deviation.json
    has the longer clips and the similarity scores)."""
    from oracle import chroma as oc, cqt as od, cqt_recursive as orc
    from tools import signals
    y = signals.polyphonic_clip(1.5, seed=100)
    a, b = np.abs(od.cqt(y)), np.abs(orc.cqt(y))
    assert a.shape == b.shape == (84, 1 + len(y) // 512)
    assert np.abs(a - b).max() < 0.01 * a.max() and np.linalg.norm(a - b) < 0.01 * np.linalg.norm(a)
    # a pure tone lands in its own bin in both
    t = np.arange(44100) / 44100
    tone = (0.5 * np.sin(2 * np.pi * 440.0 * t)).astype(np.float32)
    assert np.argmax(np.abs(orc.cqt(tone))[:, 40]) == np.argmax(np.abs(od.cqt(tone))[:, 40]) == 45      # A4 = C1 + 45 semitones
    ca, cb = oc.chroma_cqt(y), orc.chroma_cqt(y)
    assert np.abs(ca - cb).max() < 0.02 and oc.cosine(ca, cb) > 0.9999
