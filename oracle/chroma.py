"""Oracle: chroma_cqt and the auto-matcher's similarity score (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates what /root/reference/aegis_engine_core/auto_matcher.py:52-83 asks of librosa 0.10:
`feature.chroma_cqt(y, sr)` = |cqt(fmin=C1, 7 octaves x 36 bins, hop 512)| folded by `filters.cq_to_chroma`
(3 bins per semitone, centred: roll by -1; octaves tiled; rolled to start at C) and normalised per frame by its
maximum (`util.normalize(norm=inf)`, frames whose maximum is below float tiny are left as they are), and
`feature.melspectrogram(y, sr, n_mels=128)`; score = 0.4 cos(mel) + 0.6 cos(chroma) clipped to [0, 1].

`chroma_cqt` passes tuning=None, so `cqt` first estimates the tuning of the signal (`estimate_tuning`: `piptrack` peaks
above the median magnitude -> `pitch_tuning` histogram of their deviation from the 36-bins-per-octave grid, 0.01-bin
resolution) and shifts the filter bank's fmin by it; the chroma folding keeps the nominal fmin.

PARITY UNPINNED (no librosa): the CQT is the direct transform of oracle/cqt.py."""
import numpy as np

from . import cqt as ocqt
from . import dsp


def cq_to_chroma(n_input, bins_per_octave=36, n_chroma=12, fmin=32.70319566257483):
    n_merge = bins_per_octave // n_chroma
    m = np.repeat(np.eye(n_chroma), n_merge, axis=1)
    m = np.roll(m, -(n_merge // 2), axis=1)
    n_oct = int(np.ceil(n_input / bins_per_octave))
    m = np.tile(m, n_oct)[:, :n_input]
    midi_0 = np.mod(12.0 * (np.log2(fmin) - np.log2(440.0)) + 69.0, 12)
    roll = int(np.round(midi_0 * (n_chroma / 12.0)))
    return np.roll(m, roll, axis=0).astype(np.float32)


def piptrack(y, sr=44100, n_fft=2048, fmin=150.0, fmax=4000.0, threshold=0.1):
    """core/pitch.py::piptrack (hop n_fft // 4, ref = per-frame maximum) -> (pitches, mags) float32 [1 + n_fft/2, F]."""
    S = np.abs(dsp.stft(y, n_fft=n_fft, hop_length=n_fft // 4))                 # float32
    fmax = min(fmax, sr / 2.0)
    freqs = np.fft.rfftfreq(n_fft, 1.0 / sr)
    avg = np.gradient(S, axis=0)
    # util._parabolic_interpolation along the frequency axis (edges 0; 0 where |b| >= |a|)
    shift = np.zeros_like(S)
    a = S[2:] + S[:-2] - 2 * S[1:-1]
    b = (S[2:] - S[:-2]) / 2
    with np.errstate(divide="ignore", invalid="ignore"):
        inner = np.where(np.abs(b) >= np.abs(a), 0.0, -b / a)
    shift[1:-1] = inner
    dskew = 0.5 * avg * shift
    mask = ((fmin <= freqs) & (freqs < fmax)).reshape(-1, 1)
    ref = threshold * np.max(S, axis=0, keepdims=True)
    x = S * (S > ref)
    xp = np.pad(x, ((1, 1), (0, 0)), mode="edge")
    localmax = (x > xp[:-2]) & (x >= xp[2:])
    idx = np.nonzero(mask & localmax)
    pitches, mags = np.zeros_like(S), np.zeros_like(S)
    pitches[idx] = (idx[0] + shift[idx]) * float(sr) / n_fft
    mags[idx] = S[idx] + dskew[idx]
    return pitches, mags


def pitch_tuning(frequencies, resolution=0.01, bins_per_octave=12):
    """core/pitch.py::pitch_tuning: histogram peak of the deviation from the equal-tempered grid, in fractions of a bin."""
    frequencies = np.atleast_1d(frequencies)
    frequencies = frequencies[frequencies > 0]
    if not frequencies.size:
        return 0.0
    residual = np.mod(bins_per_octave * np.log2(frequencies / (440.0 / 16)), 1.0)
    residual[residual >= 0.5] -= 1.0
    bins = np.linspace(-0.5, 0.5, int(np.ceil(1.0 / resolution)) + 1)
    counts, tuning = np.histogram(residual, bins)
    return float(tuning[np.argmax(counts)])


def estimate_tuning(y, sr=44100, n_fft=2048, resolution=0.01, bins_per_octave=12):
    """core/pitch.py::estimate_tuning."""
    pitch, mag = piptrack(y, sr=sr, n_fft=n_fft)
    pitch_mask = pitch > 0
    threshold = np.median(mag[pitch_mask]) if pitch_mask.any() else 0.0
    return pitch_tuning(pitch[(mag >= threshold) & pitch_mask], resolution=resolution, bins_per_octave=bins_per_octave)


def chroma_cqt(y, sr=44100, hop_length=512, n_chroma=12, n_octaves=7, bins_per_octave=36, tuning=None):
    fmin = 32.70319566257483
    if tuning is None:
        tuning = estimate_tuning(y, sr=sr, bins_per_octave=bins_per_octave)
    C = np.abs(ocqt.cqt(y, sr=sr, hop_length=hop_length, n_bins=n_octaves * bins_per_octave,
                        fmin=fmin * 2.0 ** (tuning / bins_per_octave),
                        bins_per_octave=bins_per_octave)).astype(np.float32)
    chroma = cq_to_chroma(C.shape[0], bins_per_octave, n_chroma) @ C
    mag = np.max(np.abs(chroma), axis=0, keepdims=True)
    mag = np.where(mag < np.finfo(chroma.dtype).tiny, 1.0, mag)
    return chroma / mag


def cosine(a, b):
    a, b = a.ravel(), b.ravel()
    return float(np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-8))


def similarity(y_orig, y_synth, sr=44100):
    n = min(len(y_orig), len(y_synth))
    y_orig, y_synth = y_orig[:n], y_synth[:n]
    if n < sr * 0.5:
        return 0.0
    mel_o, mel_s = dsp.melspectrogram(y_orig, sr=sr), dsp.melspectrogram(y_synth, sr=sr)
    ch_o, ch_s = chroma_cqt(y_orig, sr), chroma_cqt(y_synth, sr)
    score = 0.4 * cosine(mel_o, mel_s) + 0.6 * cosine(ch_o, ch_s)
    return max(0.0, min(1.0, score))
