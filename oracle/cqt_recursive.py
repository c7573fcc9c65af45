"""Oracle variant: librosa's OWN way of computing the constant-Q transform (TEST INFRASTRUCTURE, see oracle/__init__.py).

`librosa.cqt` / `vqt` (core/constantq.py, 0.10) does not evaluate the direct transform of oracle/cqt.py: it works octave
by octave from the top -- the top `bins_per_octave` filters are applied as a sparsified FFT-domain basis
(`__vqt_filter_fft`: wavelets zero-padded to a power of two, transformed, each row cut at the `sparsity = 0.01` quantile
of its magnitude mass) to a rectangular-window STFT of the signal (`__cqt_response`), then the signal is resampled by 2
(`res_type="soxr_hq"`, energy preserving), the hop halved, and the same filters -- an octave lower in Hz -- are applied
again.  This file restates that recursion so that the distance between librosa's number and the direct transform the
GPU computes can be bounded (tools/cqt_deviation.py).  The one stand-in: libsoxr is absent, the
halving resampler is scipy's polyphase FIR (`scipy.signal.resample_poly(y, 1, 2)`, librosa's own `res_type="polyphase"`);
soxr_hq's pass band is flatter, so the deviation measured with the stand-in bounds librosa's from above in the top
decade of each octave and is of the same order elsewhere.  PARITY UNPINNED (no librosa here)."""
import numpy as np
import scipy.signal

from . import cqt as direct


def _wavelet(freqs, sr, alpha, filter_scale=1.0):
    """filters.wavelet(norm=1, window='hann', pad_fft=True, gamma=0): complex64 [n, n_fft], lengths."""
    lengths = filter_scale * sr / (freqs * alpha)
    n_fft = int(2.0 ** np.ceil(np.log2(lengths.max())))
    out = np.zeros((len(freqs), n_fft), dtype=np.complex64)
    for i, (f, ilen) in enumerate(zip(freqs, lengths)):
        m = np.arange(-ilen // 2, ilen // 2, dtype=float)
        L = len(m)
        win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(L) / L)
        sig = np.exp(1j * m * 2 * np.pi * f / sr) * win
        sig = sig / np.sum(np.abs(sig))
        lpad = (n_fft - L) // 2                      # util.pad_center
        out[i, lpad:lpad + L] = sig
    return out, lengths


def _sparsify_rows(x, quantile=0.01):
    """util.sparsify_rows: zero the smallest entries of each row that carry less than `quantile` of its magnitude sum."""
    x = x.copy()
    mags = np.abs(x)
    norms = np.sum(mags, axis=1, keepdims=True)
    mag_sort = np.sort(mags, axis=1)
    cum = np.cumsum(mag_sort / norms, axis=1)
    idx = np.argmin(cum < quantile, axis=1)
    for i, j in enumerate(idx):
        x[i, mags[i] < mag_sort[i, j]] = 0
    return x


def _stft_ones(y, n_fft, hop):
    """stft(window='ones', center=True, pad_mode='constant') -> complex64 [1 + n_fft/2, 1 + len(y)//hop]."""
    yp = np.pad(np.asarray(y, dtype=np.float32), n_fft // 2)
    F = 1 + (len(yp) - n_fft) // hop
    idx = np.arange(n_fft)[:, None] + hop * np.arange(F)[None, :]
    return np.fft.rfft(yp[idx].astype(np.float64), axis=0).astype(np.complex64)


def cqt(y, sr=44100, hop_length=512, n_bins=84, fmin=32.70319566257483, bins_per_octave=12, filter_scale=1.0, sparsity=0.01,
        resampler=None):
    """librosa.cqt(y, sr, hop_length, fmin, n_bins, bins_per_octave, filter_scale, norm=1, sparsity, window='hann',
    scale=True, pad_mode='constant') by its own recursion -> complex64 [n_bins, 1 + len(y)//hop_length]."""
    n_octaves = int(np.ceil(n_bins / bins_per_octave))
    n_filters = min(bins_per_octave, n_bins)
    if hop_length % (2 ** (n_octaves - 1)):
        raise ValueError("hop_length must be a multiple of 2^(n_octaves - 1)")
    freqs = direct.cqt_frequencies(n_bins, fmin, bins_per_octave)
    r = 2.0 ** (1.0 / bins_per_octave)
    alpha = (r ** 2 - 1) / (r ** 2 + 1)
    resampler = resampler or (lambda x: scipy.signal.resample_poly(x, 1, 2).astype(np.float32) * np.float32(np.sqrt(2.0)))
    my_y, my_sr, my_hop = np.asarray(y, dtype=np.float32), float(sr), hop_length
    resp = []
    for i in range(n_octaves):
        sl = slice(-n_filters, None) if i == 0 else slice(-n_filters * (i + 1), -n_filters * i)
        basis, lengths = _wavelet(freqs[sl], my_sr, alpha, filter_scale)
        n_fft = basis.shape[1]
        basis = basis * (lengths[:, None] / float(n_fft))
        fft_basis = _sparsify_rows(np.fft.fft(basis, n=n_fft, axis=1)[:, :n_fft // 2 + 1], sparsity)
        fft_basis = fft_basis * np.sqrt(sr / my_sr)
        resp.append(fft_basis.dot(_stft_ones(my_y, n_fft, my_hop)))
        if my_hop % 2 == 0:
            my_hop //= 2
            my_sr /= 2.0
            my_y = resampler(my_y)
    # __trim_stack: octaves from the top down, clipped to the shortest, the lowest n_bins rows kept
    max_col = min(c.shape[-1] for c in resp)
    V = np.zeros((n_bins, max_col), dtype=np.complex64)
    end = n_bins
    for c in resp:
        n_oct = c.shape[0]
        if end < n_oct:
            V[:end] = c[-end:, :max_col]
        else:
            V[end - n_oct:end] = c[:, :max_col]
        end -= n_oct
    lengths = filter_scale * sr / (freqs * alpha)
    return V / np.sqrt(lengths)[:, None]


def chroma_cqt(y, sr=44100, hop_length=512, n_chroma=12, n_octaves=7, bins_per_octave=36, tuning=None):
    """feature.chroma_cqt over the recursive transform (oracle/chroma.py::chroma_cqt uses the direct one)."""
    from . import chroma
    fmin = 32.70319566257483
    if tuning is None:
        tuning = chroma.estimate_tuning(y, sr=sr, bins_per_octave=bins_per_octave)
    C = np.abs(cqt(y, sr=sr, hop_length=hop_length, n_bins=n_octaves * bins_per_octave,
                   fmin=fmin * 2.0 ** (tuning / bins_per_octave), bins_per_octave=bins_per_octave)).astype(np.float32)
    ch = chroma.cq_to_chroma(C.shape[0], bins_per_octave, n_chroma) @ C
    mag = np.max(np.abs(ch), axis=0, keepdims=True)
    mag = np.where(mag < np.finfo(ch.dtype).tiny, 1.0, mag)
    return ch / mag
