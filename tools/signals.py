"""Seeded synthetic clips shaped like the reference's own fixtures (no datasets ship).

* `karplus_strong`, `noise_rake`, `guitar_test_track`: the recipe of
  /root/reference/generate_test_signal.py:5-89 (Karplus-Strong E2/A2/D3 plucks with two 25 ms
  noise rakes), re-created with an explicit seed -- the reference's generator is unseeded.
* `sine_sweep`: BASELINE.json configs[0] "10 s mono sine sweep".
* `guitar_clip`: SURVEY.md section 8d config 2/4 -- random KS notes MIDI 40..84, occasional
  rake bursts, -30 dBFS noise floor.
* `polyphonic_clip`: config 3 -- three KS voices summed.
* `c_major_scale`: /root/reference/benchmark_aegis.py:17-47 (sine scale + noise burst + hiss).
"""
import numpy as np
from scipy.signal import lfilter


def karplus_strong(frequency, duration, sr=44100, decay_factor=0.996, rng=None):
    """Plucked string; same recurrence as generate_test_signal.py:5-42, evaluated one string
    period at a time: new[p] = a*(old[p] + new[p-1]), new[0] = a*(old[0] + old[N-1])."""
    rng = np.random.default_rng(0) if rng is None else rng
    N = int(sr / frequency)
    n_samples = int(sr * duration)
    buf = rng.uniform(-1, 1, N)
    a = 0.5 * decay_factor
    out = np.empty(n_samples)
    pos = 0
    while pos < n_samples:
        take = min(N, n_samples - pos)
        out[pos:pos + take] = buf[:take]
        pos += take
        buf, _ = lfilter([a], [1.0, -a], buf, zi=[a * buf[-1]])
    return out


def noise_rake(duration, sr=44100, rng=None):
    """generate_test_signal.py:44-53: N(0, 0.8) burst with a linspace(1,0)**2 envelope."""
    rng = np.random.default_rng(0) if rng is None else rng
    n = int(sr * duration)
    return rng.normal(0, 0.8, n) * np.linspace(1, 0, n) ** 2


def guitar_test_track(sr=44100, seed=20260220):
    """generate_test_signal.py:55-97 -> float32, N = 184014 samples (~4.17 s) at 44.1 kHz."""
    rng = np.random.default_rng(seed)
    e2 = karplus_strong(82.41, 1.0, sr, rng=rng)
    a2 = karplus_strong(110.00, 1.0, sr, rng=rng)
    rake = noise_rake(0.025, sr, rng=rng)
    d3 = karplus_strong(146.83, 1.5, sr, rng=rng)
    silence = np.zeros(int(0.2 * sr))
    track = np.concatenate([silence, e2, silence, rake, silence[:1000], a2, silence, rake, d3])
    track = track / np.max(np.abs(track)) * 0.9
    return track.astype(np.float32)


def sine_sweep(duration=10.0, sr=44100, f_start=82.4068892282175, f_end=1046.5022612023945, amplitude=0.5):
    """Exponential sweep E2 -> C6."""
    n = int(sr * duration)
    t = np.arange(n) / sr
    k = np.log(f_end / f_start) / duration
    phase = 2 * np.pi * f_start * (np.exp(k * t) - 1) / k
    return (amplitude * np.sin(phase)).astype(np.float32)


def guitar_clip(duration, sr=44100, seed=1, noise_dbfs=-30.0):
    """Concatenated KS notes (MIDI 40..84, 0.15..0.8 s), 5 % chance of a 25 ms rake between
    notes, white noise floor; peak-normalised to 0.9."""
    rng = np.random.default_rng(seed)
    n_total = int(sr * duration)
    parts, n = [], 0
    while n < n_total:
        midi = int(rng.integers(40, 85))
        dur = float(rng.uniform(0.15, 0.8))
        note = karplus_strong(440.0 * 2 ** ((midi - 69) / 12), dur, sr, rng=rng)
        parts.append(note)
        n += len(note)
        if rng.random() < 0.05:
            r = noise_rake(0.025, sr, rng=rng)
            parts.append(r)
            n += len(r)
    y = np.concatenate(parts)[:n_total]
    y = y / np.max(np.abs(y)) * 0.9
    y = y + rng.normal(0, 10 ** (noise_dbfs / 20), n_total)
    return y.astype(np.float32)


def polyphonic_clip(duration, sr=44100, seed=100, voices=3):
    rng = np.random.default_rng(seed)
    y = np.zeros(int(sr * duration))
    for v in range(voices):
        y += guitar_clip(duration, sr, seed=int(rng.integers(1 << 30)), noise_dbfs=-60.0)
    return (y / np.max(np.abs(y)) * 0.9).astype(np.float32)


def c_major_scale(sr=22050, seed=7):
    """benchmark_aegis.py:17-47: eight 0.5 s sines, a 50 ms N(0,0.8) burst at 1.0 s, hiss."""
    rng = np.random.default_rng(seed)
    y = np.array([])
    for n in [60, 62, 64, 65, 67, 69, 71, 72]:
        f = 440.0 * 2 ** ((n - 69) / 12)
        t = np.linspace(0, 0.5, int(sr * 0.5))
        y = np.concatenate([y, 0.5 * np.sin(2 * np.pi * f * t)])
    s, d = int(sr * 1.0), int(sr * 0.05)
    y[s:s + d] += rng.normal(0, 0.8, d)
    y += rng.normal(0, 0.02, len(y))
    return y.astype(np.float32)


def pitched_start_clip(sr=44100):
    """A clip that is pitched from its very first sample (880 Hz sine at phase 0, a pause, a 220 Hz sine): frame 0 is
    where the initial distribution of pYIN's HMM decides -- librosa's unvoiced start leaves it unvoiced, the uniform
    start voices it (tests of both modes)."""
    t = np.arange(int(0.8 * sr)) / sr
    return np.concatenate([0.5 * np.sin(2 * np.pi * 880.0 * t), np.zeros(int(0.3 * sr)),
                           0.4 * np.sin(2 * np.pi * 220.0 * t)]).astype(np.float32)


def hostile_clips(sr=44100, seconds=2.0):
    """Signals a folder of real recordings holds and a synthetic guitar does not: offsets, clipping, impulse trains, levels
    around the difference function's 1e-6 clamps, tones outside [fmin, fmax], beating, the Nyquist tone, float32 denormals,
    a step.  name -> float32 clip (seeded)."""
    n = int(seconds * sr)
    t = np.arange(n) / sr
    g = guitar_clip(seconds, sr=sr, seed=77)
    rng = np.random.default_rng(77)
    imp = np.zeros(n, np.float32)
    imp[::sr // 110] = 0.9
    step = np.zeros(n, np.float32)
    step[n // 2:] = 0.7
    out = {
        "dc_offset": g * np.float32(0.5) + np.float32(0.5),
        "clipped": np.clip(g * np.float32(8.0), -1.0, 1.0),
        "impulse_train": imp,
        "quiet_1e-3": g * np.float32(1e-3),
        "quiet_1e-5": g * np.float32(1e-5),
        "tone_3k_above_fmax": 0.5 * np.sin(2 * np.pi * 3000.0 * t),
        "tone_40_below_fmin": 0.5 * np.sin(2 * np.pi * 40.0 * t),
        "beating_220_223": 0.4 * np.sin(2 * np.pi * 220.0 * t) + 0.4 * np.sin(2 * np.pi * 223.0 * t),
        "square_100": np.where(np.sin(2 * np.pi * 100.0 * t) >= 0, 1.0, -1.0),
        "nyquist": 0.9 * np.where(np.arange(n) % 2 == 0, 1.0, -1.0),
        "denormal": np.full(n, 1e-39),
        "tremolo_440": 0.5 * np.sin(2 * np.pi * 440.0 * t) * (0.55 + 0.45 * np.sin(2 * np.pi * 8.0 * t)),
        "white_uniform": rng.uniform(-1.0, 1.0, n),
        "step": step,
        "fast_chirp": 0.5 * np.sin(2 * np.pi * (50.0 * t + 0.5 * (4000.0 - 50.0) / seconds * t * t)),
    }
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in out.items()}
