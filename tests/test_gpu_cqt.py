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
