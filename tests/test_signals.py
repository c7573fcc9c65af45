"""The synthetic fixtures follow the recipe of the reference's generator (generate_test_signal.py:5-53): the
block-wise Karplus-Strong of signals.py must satisfy the generator's sample recurrence."""
import numpy as np

from tools import signals


def test_karplus_strong_satisfies_the_sample_recurrence():
    # generate_test_signal.py:22-40 reads buf[ptr], averages it with the previous (already updated) cell and writes
    # it back scaled by the decay: y[i] = 0.5 * decay * (y[i - N] + y[i - 1]) for i >= N, the first N samples being
    # the random excitation.
    for freq, dur, decay in ((82.41, 0.5, 0.996), (440.0, 0.2, 0.99), (1046.5, 0.05, 0.996)):
        y = signals.karplus_strong(freq, dur, 44100, decay_factor=decay, rng=np.random.default_rng(3))
        N = int(44100 / freq)
        assert len(y) == int(44100 * dur)
        assert np.all(np.abs(y[:N]) <= 1.0)
        want = 0.5 * decay * (y[:-N] + y[N - 1:-1])
        np.testing.assert_allclose(y[N:], want, rtol=0, atol=1e-12)


def test_fixture_shapes():
    y = signals.guitar_test_track()
    assert y.dtype == np.float32 and len(y) == 184014
    assert abs(float(np.max(np.abs(y))) - 0.9) < 1e-6
    assert len(signals.sine_sweep()) == 441000


def test_hostile_clips_are_seeded_float32_and_finite():
    """tools/signals.py::hostile_clips (the GPU parity test on offsets, clipping, impulses, levels around the 1e-6 clamps,
    out-of-range tones, the Nyquist tone, denormals, a step): same arrays on every call, float32, finite, within [-1, 1]."""
    import numpy as np
    from tools import signals
    a, b = signals.hostile_clips(), signals.hostile_clips()
    assert len(a) == 15 and list(a) == list(b)
    for k in a:
        assert a[k].dtype == np.float32 and a[k].shape == (88200,) and np.isfinite(a[k]).all() and np.abs(a[k]).max() <= 1.0 + 1e-6, k
        assert np.array_equal(a[k], b[k]), k
    assert a["denormal"].max() < 1.2e-38 and a["denormal"].min() > 0                 # float32 denormals, not zeros
    assert np.abs(a["clipped"]).max() == 1.0 and (np.abs(a["clipped"]) == 1.0).mean() > 0.01
