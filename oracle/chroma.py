"""Oracle: chroma_cqt and the auto-matcher's similarity score (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates what /root/reference/aegis_engine_core/auto_matcher.py:52-83 asks of librosa 0.10:
`feature.chroma_cqt(y, sr)` = |cqt(fmin=C1, 7 octaves x 36 bins, hop 512)| folded by `filters.cq_to_chroma`
(3 bins per semitone, centred: roll by -1; octaves tiled; rolled to start at C) and normalised per frame by its
maximum (`util.normalize(norm=inf)`, frames whose maximum is below float tiny are left as they are), and
`feature.melspectrogram(y, sr, n_mels=128)`; score = 0.4 cos(mel) + 0.6 cos(chroma) clipped to [0, 1].

PARITY UNPINNED (no librosa): the CQT is the direct transform of oracle/cqt.py, and `tuning` is fixed at 0.0 where
librosa estimates it from the signal (`estimate_tuning`, a piptrack histogram) before building the filters."""
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


def chroma_cqt(y, sr=44100, hop_length=512, n_chroma=12, n_octaves=7, bins_per_octave=36):
    C = np.abs(ocqt.cqt(y, sr=sr, hop_length=hop_length, n_bins=n_octaves * bins_per_octave,
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
