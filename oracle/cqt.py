"""Oracle: direct constant-Q transform (TEST INFRASTRUCTURE, see oracle/__init__.py).

The only CQT call in the reference is `librosa.feature.chroma_cqt(y, sr)` inside the auto-matcher's similarity
score (/root/reference/aegis_engine_core/auto_matcher.py:68-69); BASELINE.json configs[2] asks for an 84-bin
filter bank.  librosa.cqt evaluates the transform octave by octave on resampled (soxr) copies of the signal --
an approximation of the direct transform restated here from librosa 0.10 `filters.wavelet` / `core/constantq.py`:

    atom_k[m] = hann_periodic(L_k)[m - lo_k] * exp(2 pi i f_k m / sr) / sum(hann),  m in [floor(-N_k/2), floor(N_k/2))
    N_k = filter_scale * sr / (f_k * alpha),  alpha = (r^2 - 1) / (r^2 + 1),  r = 2^(1/bins_per_octave)
    C[k, t] = sqrt(N_k) * sum_m atom_k[m] * y_pad[t*hop - m]        (scale=True, norm=1, center=True, zero padding)

PARITY UNPINNED: no librosa here, and librosa's own result differs from this by its resampling error (~1e-3)."""
import numpy as np
import scipy.signal


def cqt_frequencies(n_bins=84, fmin=32.70319566257483, bins_per_octave=12):
    return fmin * 2.0 ** (np.arange(n_bins) / bins_per_octave)


def wavelet_lengths(freqs, sr, bins_per_octave=12, filter_scale=1.0):
    r = 2.0 ** (1.0 / bins_per_octave)
    alpha = (r ** 2 - 1) / (r ** 2 + 1)
    return (filter_scale / alpha) * sr / freqs


def atoms(sr=44100, n_bins=84, fmin=32.70319566257483, bins_per_octave=12, filter_scale=1.0):
    """-> list of (m_lo, complex atom over m = m_lo .. m_lo+L-1, N_k)."""
    freqs = cqt_frequencies(n_bins, fmin, bins_per_octave)
    out = []
    for f, ilen in zip(freqs, wavelet_lengths(freqs, sr, bins_per_octave, filter_scale)):
        m = np.arange(-ilen // 2, ilen // 2, dtype=float)
        L = len(m)
        win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(L) / L)
        sig = np.exp(1j * m * 2 * np.pi * f / sr) * win
        sig = sig / np.sum(np.abs(sig))
        out.append((int(m[0]), sig, ilen))
    return out


def cqt(y, sr=44100, hop_length=512, n_bins=84, fmin=32.70319566257483, bins_per_octave=12, filter_scale=1.0):
    """complex128 [n_bins, 1 + len(y)//hop]."""
    y = np.asarray(y, dtype=np.float64)
    F = 1 + len(y) // hop_length
    C = np.zeros((n_bins, F), dtype=np.complex128)
    centres = np.arange(F) * hop_length
    for k, (m_lo, sig, ilen) in enumerate(atoms(sr, n_bins, fmin, bins_per_octave, filter_scale)):
        # sum_m sig[m] y[c - m]  = (y * sig)[c], a plain convolution sampled at the frame centres
        full = scipy.signal.fftconvolve(y, sig)              # index i <-> c = i + m_lo
        idx = centres - m_lo
        ok = (idx >= 0) & (idx < len(full))
        C[k, ok] = full[idx[ok]]
        C[k] *= np.sqrt(ilen)
    return C
