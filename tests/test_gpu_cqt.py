"""Constant-Q magnitudes on the MFMA units (aegis_cqt, SURVEY 8a row a19 / BASELINE configs[2]) against the
float64 direct-transform oracle (oracle/cqt.py).  Tolerance 1e-4 of the clip maximum (float32 GEMM over up to
23 372 taps).  Parity unpinned: no librosa, and librosa.cqt itself only approximates this transform."""
import numpy as np
import pytest

from oracle import cqt as ocqt
from spectrogram_midi_amd import _lib
from tools import signals

pytestmark = pytest.mark.gpu


def check(got, y, **kw):
    ref = np.abs(ocqt.cqt(y, **kw))
    assert got.shape == ref.shape and got.dtype == np.float32
    tol = 1e-4 * max(ref.max(), 1e-12)
    assert np.abs(got - ref).max() <= tol, (np.abs(got - ref).max(), ref.max())


def test_cqt_matches_direct_transform():
    h = _lib.Handle()
    t = np.arange(2 * 44100) / 44100
    clips = [(0.5 * np.sin(2 * np.pi * 220 * t)).astype(np.float32), signals.polyphonic_clip(3.0, seed=100),
             signals.guitar_test_track(), np.zeros(0, np.float32), np.zeros(300, np.float32) + 0.25]
    got = h.cqt(clips)
    for g, y in zip(got, clips):
        check(g, y)
    k = int(np.argmax(got[0][:, 80]))
    assert abs(ocqt.cqt_frequencies()[k] - 220.0) < 1e-6                # A3 lands in bin 33
    np.testing.assert_array_equal(h.cqt([clips[1]])[0], got[1])         # ragged batching changes nothing
    h.close()


def test_other_banks():
    h = _lib.Handle()
    y = signals.guitar_clip(2.0, seed=3)
    check(h.cqt([y], n_bins=36, bins_per_octave=12, fmin=65.40639132514966)[0], y, n_bins=36, fmin=65.40639132514966)
    check(h.cqt([y], n_bins=96, bins_per_octave=24, fmin=110.0)[0], y, n_bins=96, bins_per_octave=24, fmin=110.0)
    with pytest.raises(_lib.AegisError):
        h.cqt([y], n_bins=200)
    with pytest.raises(_lib.AegisError):
        h.cqt([y], n_bins=84, fmin=4000.0)                               # top bins above Nyquist
    h.close()


def test_other_hops():
    """hop 256 stays on the sliding-window kernel; hop 1024 (window too wide for the LDS ring) and hop 441
    (not a multiple of 16) take the per-frame staging kernel.  Same bank, same answer."""
    y = signals.polyphonic_clip(2.5, seed=101)
    for hop in (256, 1024, 441):
        h = _lib.Handle(hop_length=hop, scipy_tables=False)
        got = h.cqt([y, y[:30000]])
        check(got[0], y, hop_length=hop)
        check(got[1], y[:30000], hop_length=hop)
        h.close()


def test_chroma_cqt_and_similarity(tmp_path):
    """chroma_cqt (252-bin CQT on the GPU + host folding) and the auto-matcher's score against oracle/chroma.py
    (auto_matcher.py:52-83).  The score tolerance covers float32 GEMM error and the -80 dB clamp of S_dB."""
    from oracle import chroma as ochroma
    from spectrogram_midi_amd import audio_io, similarity
    h = _lib.Handle()
    a = signals.c_major_scale(sr=44100)[:3 * 44100]
    b = signals.polyphonic_clip(3.0, seed=7)
    got = similarity.chroma_cqt(h, [a, b])
    for g, y in zip(got, (a, b)):
        ref = ochroma.chroma_cqt(y)
        assert g.shape == ref.shape == (12, 1 + len(y) // 512) and g.dtype == np.float32
        assert np.abs(g - ref).max() < 2e-4 and np.abs(g.max(axis=0) - 1.0).max() < 1e-6
    # the scale's first note is C4: pitch class 0 dominates its frames
    assert int(np.argmax(got[0][:, 20])) == 0
    for x, y in ((a, a), (a, b), (a, 0.3 * a + 0.05 * b[:len(a)])):
        s = similarity.similarity_arrays(h, x, y)
        assert abs(s - ochroma.similarity(x, y)) < 2e-4, (s, ochroma.similarity(x, y))
    assert abs(similarity.similarity_arrays(h, a, a) - 1.0) < 1e-6
    assert similarity.similarity_arrays(h, a[:20000], a[:20000]) == 0.0          # < 0.5 s
    # file-level mirror of _calculate_similarity: path + WAV bytes, catch-all -> 0.0
    p = str(tmp_path / "orig.wav")
    audio_io.write_wav(p, a, 44100)
    q = str(tmp_path / "synth.wav")
    audio_io.write_wav(q, b, 44100)
    s = similarity._calculate_similarity(p, open(q, "rb").read(), 44100, handle=h)
    a16, b16 = audio_io.read_wav(p, 44100), audio_io.read_wav(q, 44100)
    assert abs(s - ochroma.similarity(a16, b16)) < 2e-4
    assert similarity._calculate_similarity(str(tmp_path / "missing.wav"), b"", 44100, handle=h) == 0.0
    h.close()


def _tuning_from_one_frame(y, sr=44100, n_fft=2048, bpo=36, start=20480):
    """What librosa.estimate_tuning must answer for a STATIONARY tone, derived from the peak positions of ONE interior
    frame and nothing else: Hann-windowed magnitude spectrum, local maxima above a tenth of the largest inside
    150..4000 Hz, piptrack's parabolic interpolation, the peaks at or above the median magnitude, their residuals in
    1/bpo-octave bins folded to [-0.5, 0.5), the fullest 0.01-wide histogram bin."""
    w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)
    S = np.abs(np.fft.rfft(y[start:start + n_fft].astype(np.float64) * w))
    fr = np.arange(len(S)) * sr / n_fft
    k = np.flatnonzero((S[1:-1] > S[:-2]) & (S[1:-1] >= S[2:]) & (S[1:-1] > 0.1 * S.max()) & (fr[1:-1] >= 150) & (fr[1:-1] <= 4000)) + 1
    avg, den = 0.5 * (S[k + 1] - S[k - 1]), 2 * S[k] - S[k - 1] - S[k + 1]
    shift = avg / den
    pitch, mag = (k + shift) * sr / n_fft, S[k] + 0.5 * avg * shift
    res = np.mod(bpo * np.log2(pitch[mag >= np.median(mag)] / (440.0 / 16)), 1.0)
    res[res >= 0.5] -= 1.0
    edges = np.linspace(-0.5, 0.5, 101)
    return edges[np.argmax(np.histogram(res, edges)[0])], res


def test_chroma_follows_the_estimated_tuning():
    """auto_matcher.py:68-69 leaves tuning=None: librosa estimates it and shifts the filter bank.  A clip played
    20 cents flat must be analysed with the shifted bank (same chroma as the oracle, different from the nominal grid).
    The expected estimate does not come from the code under test: 20 cents flat = -0.6 of a 1/36-octave bin, which
    folds to +0.4; the parabolic peak interpolation of a 2048-point Hann spectrum is good to a few hundredths of a bin
    there, and the single-frame derivation above (peak positions only) must give the very bin the estimators return."""
    from oracle import chroma as ochroma
    from spectrogram_midi_amd import similarity
    h = _lib.Handle()
    t = np.arange(3 * 44100) / 44100
    f = 261.6255653005986 * 2 ** (-20 / 1200)
    y = sum(a * np.sin(2 * np.pi * k * f * t) for k, a in ((1, 0.4), (2, 0.2), (4, 0.1))).astype(np.float32)
    expected, residuals = _tuning_from_one_frame(y)
    assert abs(expected - 0.4) <= 0.05 and np.ptp(residuals) < 0.01          # both kept partials agree: no tie to break
    tn = similarity.estimate_tuning(y, 44100, 36)
    assert tn == expected
    assert tn == ochroma.estimate_tuning(y, 44100, bins_per_octave=36)
    got = similarity.chroma_cqt(h, [y])[0]
    ref = ochroma.chroma_cqt(y)
    assert np.abs(got - ref).max() < 2e-4
    nominal = similarity.chroma_cqt(h, [y], tuning=0.0)[0]
    assert np.abs(nominal - ochroma.chroma_cqt(y, tuning=0.0)).max() < 2e-4
    assert np.abs(nominal - got).max() > 1e-2                       # the shift is visible in the chroma
    h.close()


def test_cqt_configs2_size_properties():
    """BASELINE.json configs[2] at full size: 64 x 30 s polyphonic clips, 84 bins.  Size-independent properties:
    a clip's result does not depend on the batch around it, magnitudes scale linearly with the gain, a delay of
    whole hops shifts the frames, and spot checks against the float64 oracle on two clips."""
    h = _lib.Handle()
    base = [signals.polyphonic_clip(30.0, seed=100 + i) for i in range(4)]
    rng = np.random.default_rng(3)
    clips = [base[i % 4] if i < 4 else np.roll(base[i % 4], int(rng.integers(1, 10000))) for i in range(64)]
    out = h.cqt(clips)
    assert len(out) == 64 and all(o.shape == (84, 2584) for o in out)
    np.testing.assert_array_equal(h.cqt([clips[5]])[0], out[5])                     # batch independence
    half = h.cqt([np.float32(0.5) * clips[0]])[0]
    np.testing.assert_allclose(half, 0.5 * out[0], rtol=2e-6, atol=1e-7 * out[0].max())   # linearity in the gain
    delayed = h.cqt([np.concatenate([np.zeros(8 * 512, np.float32), clips[1]])])[0]
    # interior frames (beyond the longest atom's half support, 11 686 samples = 23 frames) just move by 8 frames
    np.testing.assert_allclose(delayed[:, 8 + 24:2584 - 24], out[1][:, 24:2584 - 8 - 24], rtol=0, atol=2e-5 * out[1].max())
    ref = np.abs(ocqt.cqt(clips[2][:200 * 512], n_bins=84))
    sub = h.cqt([clips[2][:200 * 512]])[0]
    assert np.abs(sub - ref).max() <= 1e-4 * ref.max()
    h.close()


def test_device_entry_equals_the_host_entry():
    """aegis_cqt_device (PCM and magnitudes resident in HBM) == aegis_cqt, bit for bit, for a ragged batch."""
    import torch
    h = _lib.Handle()
    clips = [signals.polyphonic_clip(2.0, seed=100), signals.guitar_clip(1.3, seed=3), np.zeros(700, np.float32) + 0.1]
    want = h.cqt(clips)
    off = np.concatenate([[0], np.cumsum([len(c) for c in clips])]).astype(np.int64)
    dev = torch.device("cuda", 0)
    d_pcm = torch.from_numpy(np.concatenate(clips)).to(dev)
    F = [1 + len(c) // 512 for c in clips]
    d_out = torch.zeros(sum(F) * 84, dtype=torch.float32, device=dev)
    h.cqt_device(d_pcm.data_ptr(), off, d_out.data_ptr())
    got = d_out.cpu().numpy()
    o = 0
    for Fc, w in zip(F, want):
        np.testing.assert_array_equal(got[o:o + Fc * 84].reshape(84, Fc), w)
        o += Fc * 84
    h.close()


def test_chroma_folded_on_the_device_equals_the_host_fold():
    """aegis_chroma_cqt (folding + per-frame max normalisation behind the magnitudes, on the device) against the same two
    steps done by NumPy on the magnitudes aegis_cqt returns: same bins added, float32, so equal to rounding -- on a ragged
    batch with an empty clip, a silent one (frames left un-normalised) and one shorter than a hop."""
    from spectrogram_midi_amd import similarity
    h = _lib.Handle()
    clips = [signals.polyphonic_clip(2.0, seed=3), np.zeros(0, np.float32), np.zeros(30000, np.float32),
             signals.guitar_clip(1.3, seed=4)[:300], signals.guitar_clip(3.1, seed=5)]
    tunings = [0.0] * len(clips)
    fold = similarity.cq_to_chroma(252, 36, 12)
    dev = similarity.chroma_cqt(h, clips, tuning=0.0)
    host = similarity._chroma_cqt_host_fold(h, clips, tunings, fold, 252, 36, similarity._C1)
    for a, b, y in zip(dev, host, clips):
        assert a.shape == b.shape == (12, 1 + len(y) // 512) and a.dtype == np.float32
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-6)
    assert not dev[2].any() and dev[0].max() == 1.0
    with pytest.raises(ValueError):
        h.chroma_cqt(clips[:1], np.zeros(5, np.int32))
    h.close()
