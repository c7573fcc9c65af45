"""Constant-Q magnitudes on the MFMA units (aegis_cqt, SURVEY 8a row a19 / BASELINE configs[2]) against the
float64 direct-transform oracle (oracle/cqt.py).  Tolerance 1e-4 of the clip maximum (float32 GEMM over up to
23 372 taps).  Parity unpinned: no librosa, and librosa.cqt itself only approximates this transform."""
import numpy as np
import pytest

from oracle import cqt as ocqt
from spectrogram_midi_amd import _lib, signals

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
