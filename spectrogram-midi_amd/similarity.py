"""Spectral similarity score of the auto-matcher on the GPU path.

Mirror of `_calculate_similarity` (reference aegis_engine_core/auto_matcher.py:13-89): 0.4 x cosine of the two
128-band mel power spectrograms + 0.6 x cosine of the two `chroma_cqt` matrices, clipped to [0, 1]; inputs shorter
than 0.5 s score 0.  The mel spectrograms come from the analyze path (stage MEL) and the 252-bin constant-Q
magnitudes from `aegis_chroma_cqt` (7 octaves x 36 bins from C1 on the MFMA units, folded into the 12 chroma classes
and normalised per frame on the device); only the two dot products run on the host.

`chroma_cqt` leaves `tuning=None`, so librosa first estimates each signal's tuning (a `piptrack` histogram over the
36-bins-per-octave grid) and shifts the filter bank by it; `estimate_tuning` below does the same on the host (one STFT of
the clip), and the bank is built for the shifted fmin.

Differences a maintainer should know (all below the score's third decimal on the test signals): the CQT is the
direct transform (librosa's multirate recursion approximates it), and the mel cosine is taken on 10^(S_dB/10) -- proportional to the mel power except where it sits more
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


def _stft_magnitude(y, n_fft=2048, hop_length=512):
    """|librosa.stft(y)| (center=True, zero padding, periodic Hann): float32 [1 + n_fft/2, 1 + len(y)//hop]."""
    y = np.asarray(y, dtype=np.float32)
    ypad = np.concatenate([np.zeros(n_fft // 2, np.float32), y, np.zeros(n_fft // 2, np.float32)])
    n_frames = 1 + (len(ypad) - n_fft) // hop_length
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n_fft) / n_fft)
    out = np.empty((1 + n_fft // 2, n_frames), np.float32)
    for s0 in range(0, n_frames, 1024):                       # blocks keep the float64 work array small
        idx = (np.arange(s0, min(s0 + 1024, n_frames)) * hop_length)[:, None] + np.arange(n_fft)[None, :]
        out[:, s0:s0 + idx.shape[0]] = np.abs(np.fft.rfft(ypad[idx] * win, axis=1).astype(np.complex64)).T
    return out


def estimate_tuning(y, sr=44100, bins_per_octave=36, n_fft=2048, resolution=0.01):
    """librosa.estimate_tuning(y=y, sr=sr, bins_per_octave=...) as `cqt(tuning=None)` calls it: spectral peaks between
    150 and 4000 Hz (`piptrack`: local maxima above a tenth of the frame maximum, refined by parabolic interpolation),
    those at or above the median peak magnitude, and the most populated 0.01-bin cell of their deviation from the
    equal-tempered grid.  Returns the deviation in fractions of a bin, in [-0.5, 0.5)."""
    S = _stft_magnitude(y, n_fft, n_fft // 4)
    freqs = np.fft.rfftfreq(n_fft, 1.0 / sr)
    lo, hi = S[:-2], S[2:]
    mid = S[1:-1]
    a = hi + lo - 2 * mid
    b = (hi - lo) / 2
    with np.errstate(divide="ignore", invalid="ignore"):
        shift = np.where(np.abs(b) >= np.abs(a), np.float32(0), -b / a)
    gate = S * (S > 0.1 * S.max(axis=0, keepdims=True))
    peak = (gate[1:-1] > gate[:-2]) & (gate[1:-1] >= gate[2:])
    peak &= ((freqs[1:-1] >= 150.0) & (freqs[1:-1] < min(4000.0, sr / 2.0)))[:, None]
    k, t = np.nonzero(peak)
    if not len(k):
        return 0.0
    pitch = ((k + 1 + shift[k, t]) * float(sr) / n_fft).astype(np.float32)
    mag = mid[k, t] + 0.5 * b[k, t] * shift[k, t]
    keep = (pitch > 0)
    pitch, mag = pitch[keep], mag[keep]
    if not len(pitch):
        return 0.0
    pitch = pitch[mag >= np.median(mag)]
    residual = np.mod(bins_per_octave * np.log2(pitch / np.float32(440.0 / 16)), np.float32(1.0))
    residual[residual >= 0.5] -= 1.0
    edges = np.linspace(-0.5, 0.5, int(np.ceil(1.0 / resolution)) + 1)
    counts, _ = np.histogram(residual, edges)
    return float(edges[np.argmax(counts)])


def chroma_cqt(handle, clips, n_chroma=12, n_octaves=7, bins_per_octave=36, fmin=_C1, tuning=None):
    """librosa.feature.chroma_cqt(y, sr) for every clip: float32 [n_chroma, F], each frame divided by its maximum.
    `tuning=None` estimates it per clip as librosa does; a number fixes it (0.0 = the nominal grid)."""
    n_bins = n_octaves * bins_per_octave
    tunings = [estimate_tuning(c, handle.sr, bins_per_octave) if tuning is None else float(tuning) for c in clips]
    fold = cq_to_chroma(n_bins, bins_per_octave, n_chroma, fmin)
    if n_chroma > 24 or not np.array_equal(fold.sum(axis=0), np.ones(n_bins, np.float32)):
        return _chroma_cqt_host_fold(handle, clips, tunings, fold, n_bins, bins_per_octave, fmin)
    # every CQT bin feeds exactly one chroma class: folding and the per-frame max normalisation run on the device behind
    # the magnitudes (aegis_chroma_cqt), and n_chroma x F floats come back instead of n_bins x F
    cls = np.argmax(fold, axis=0).astype(np.int32)
    out = [None] * len(clips)
    for tn in sorted(set(tunings)):                           # one filter bank per distinct tuning
        sel = [i for i, x in enumerate(tunings) if x == tn]
        res = handle.chroma_cqt([clips[i] for i in sel], cls, n_chroma=n_chroma, n_bins=n_bins, bins_per_octave=bins_per_octave,
                                fmin=fmin * 2.0 ** (tn / bins_per_octave))
        for i, ch in zip(sel, res):
            out[i] = ch
    return out


def _chroma_cqt_host_fold(handle, clips, tunings, fold, n_bins, bins_per_octave, fmin):
    """chroma_cqt with the folding matrix applied on the host (a folding matrix that is not 0/1 with one class per bin)."""
    mags = [None] * len(clips)
    for tn in sorted(set(tunings)):
        sel = [i for i, x in enumerate(tunings) if x == tn]
        res = handle.cqt([clips[i] for i in sel], n_bins=n_bins, bins_per_octave=bins_per_octave,
                         fmin=fmin * 2.0 ** (tn / bins_per_octave))
        for i, C in zip(sel, res):
            mags[i] = C
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
