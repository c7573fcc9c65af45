"""Spectral similarity score of the auto-matcher on the GPU path.

Mirror of `_calculate_similarity` (reference aegis_engine_core/auto_matcher.py:13-89): 0.4 x cosine of the two
128-band mel power spectrograms + 0.6 x cosine of the two `chroma_cqt` matrices, clipped to [0, 1]; inputs shorter
than 0.5 s score 0.  The mel spectrograms come from the analyze path (stage MEL) and the 252-bin constant-Q
magnitudes from `aegis_cqt` (7 octaves x 36 bins from C1 on the MFMA units); only the 12 x 252 chroma folding and
the two dot products run on the host.

Differences a maintainer should know (all below the score's third decimal on the test signals): the CQT is the
direct transform (librosa's multirate recursion approximates it), `tuning` is 0.0 where librosa estimates it from
the signal, and the mel cosine is taken on 10^(S_dB/10) -- proportional to the mel power except where it sits more
than 80 dB under the clip maximum (the cosine ignores the per-clip scale).  The grid search around this score needs
FluidSynth and stays outside this package."""
import numpy as np

from . import _lib, audio_io

_C1 = 32.70319566257483


def cq_to_chroma(n_input, bins_per_octave=36, n_chroma=12, fmin=_C1):
    """librosa.filters.cq_to_chroma(base_c=True, window=None): float32 [n_chroma, n_input]."""
    if bins_per_octave % n_chroma:
        raise ValueError("bins_per_octave must be a multiple of n_chroma")
    n_merge = bins_per_octave // n_chroma
    m = np.roll(np.repeat(np.eye(n_chroma), n_merge, axis=1), -(n_merge // 2), axis=1)
    m = np.tile(m, -(-n_input // bins_per_octave))[:, :n_input]
    midi_0 = np.mod(12.0 * (np.log2(fmin) - np.log2(440.0)) + 69.0, 12)
    return np.roll(m, int(np.round(midi_0 * (n_chroma / 12.0))), axis=0).astype(np.float32)


def chroma_cqt(handle, clips, n_chroma=12, n_octaves=7, bins_per_octave=36, fmin=_C1):
    """librosa.feature.chroma_cqt(y, sr) for every clip: float32 [n_chroma, F], each frame divided by its maximum."""
    mags = handle.cqt(clips, n_bins=n_octaves * bins_per_octave, bins_per_octave=bins_per_octave, fmin=fmin)
    fold = cq_to_chroma(n_octaves * bins_per_octave, bins_per_octave, n_chroma, fmin)
    out = []
    for C in mags:
        chroma = fold @ C
        mag = np.max(np.abs(chroma), axis=0, keepdims=True)
        out.append(chroma / np.where(mag < np.finfo(np.float32).tiny, np.float32(1.0), mag))
    return out


def _cosine(a, b):
    a, b = a.ravel(), b.ravel()
    return float(np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-8))


def similarity_arrays(handle, y_orig, y_synth):
    """The score for two signals already at the handle's sample rate."""
    n = min(len(y_orig), len(y_synth))
    if n < handle.sr * 0.5:
        return 0.0
    clips = [np.ascontiguousarray(y_orig[:n], np.float32), np.ascontiguousarray(y_synth[:n], np.float32)]
    raw = handle.analyze_batch(clips, stages=_lib.STAGE_MEL, want_sdb=True)
    mel_o, mel_s = (np.power(10.0, r["S_dB"].astype(np.float64) / 10.0) for r in raw)
    ch_o, ch_s = chroma_cqt(handle, clips)
    score = 0.4 * _cosine(mel_o, mel_s) + 0.6 * _cosine(ch_o.astype(np.float64), ch_s.astype(np.float64))
    return max(0.0, min(1.0, score))


def _calculate_similarity(original_audio_path, synthesized_wav_data, sample_rate=44100, handle=None):
    """Same arguments and error behaviour as the reference: the first 30 s of the original file against the
    synthesised WAV bytes; any failure prints and returns 0.0 (auto_matcher.py:87-89)."""
    own = handle is None
    try:
        if own:
            handle = _lib.Handle(sample_rate=sample_rate)
        y_orig = audio_io.read_wav(original_audio_path, sample_rate, duration=30)
        y_synth = audio_io.read_wav_bytes(synthesized_wav_data, sample_rate)
        return similarity_arrays(handle, y_orig, y_synth)
    except Exception as e:                                   # noqa: BLE001 -- mirrors the reference's catch-all
        print(f"[AutoMatcher] similarity failed: {e}")
        return 0.0
    finally:
        if own and handle is not None:
            handle.close()
