"""Oracle: probabilistic YIN (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates `librosa.pyin(y, fmin=E2, fmax=C6, sr, hop_length)` as the reference
calls it (`/root/reference/aegis_engine.py:63,67,190,216`,
`aegis_engine_core/worker.py:9-15`), following librosa 0.10.x
`core/pitch.py::pyin/_cumulative_mean_normalized_difference/
_parabolic_interpolation/__pyin_helper` and `sequence.py::viterbi/
transition_local/transition_loop`, with NumPy-1.x
dtype rules: float32 frames, float64 FFT-autocorrelation, float32 running
energy, float64 CMND / observation / Viterbi.  Parity unpinned.
"""
import ctypes
import os

import numpy as np
import scipy.stats

from .dsp import frame_centered

TINY = float(np.finfo(np.float64).tiny)


class PyinParams:
    def __init__(self, sr=44100, fmin=82.4068892282175, fmax=1046.5022612023945,
                 frame_length=2048, hop_length=512):
        self.sr, self.fmin, self.fmax = sr, float(fmin), float(fmax)
        self.frame_length, self.hop_length = frame_length, hop_length
        self.win_length = frame_length // 2                                   # P1
        self.min_period = int(np.floor(sr / fmax))                            # P2
        self.max_period = min(int(np.ceil(sr / fmin)), frame_length - self.win_length - 1)
        self.n_lags = self.max_period - self.min_period + 1
        self.n_thresholds = 100
        self.thresholds = np.linspace(0, 1, self.n_thresholds + 1)            # P6
        self.beta_probs = np.diff(scipy.stats.beta.cdf(self.thresholds, 2, 18))
        self.boltzmann = 2.0
        self.no_trough_prob = 0.01
        self.bins_per_semitone = int(np.ceil(1.0 / 0.1))
        self.n_pitch_bins = int(np.floor(12 * self.bins_per_semitone * np.log2(fmax / fmin))) + 1  # P9
        self.max_semitones_per_frame = round(35.92 * 12 * hop_length / sr)    # P11
        self.transition_width = self.max_semitones_per_frame * self.bins_per_semitone + 1
        self.switch_prob = 0.01
        self.freqs = self.fmin * 2 ** (np.arange(self.n_pitch_bins) / (12 * self.bins_per_semitone))  # P12


# ----------------------------------------------------------------------------
# P3 / P4: difference function and CMND
# ----------------------------------------------------------------------------
def difference_terms(y_frames, p):
    """Returns (acf float64[win+.., F] restricted to lags 0..max_period,
    energy float32[..], d float64[..]) for lags 0..max_period."""
    W, L = p.win_length, p.frame_length
    yf64 = y_frames.astype(np.float64)
    a = np.fft.rfft(yf64, L, axis=-2)
    b = np.fft.rfft(yf64[W:0:-1, :], L, axis=-2)
    acf = np.fft.irfft(a * b, L, axis=-2)[W:, :]
    acf[np.abs(acf) < 1e-6] = 0
    energy = np.cumsum(y_frames ** 2, axis=-2)          # float32, sequential along the frame
    energy = energy[W:, :] - energy[:-W, :]
    energy[np.abs(energy) < 1e-6] = 0
    d = energy[:1, :] + energy - 2 * acf
    return acf, energy, d


def cmnd(y_frames, p):
    """_cumulative_mean_normalized_difference -> float64[n_lags, F]."""
    _, _, d = difference_terms(y_frames, p)
    num = d[p.min_period : p.max_period + 1, :]
    tau = np.arange(1, p.max_period + 1).reshape(-1, 1)
    cum_mean = np.cumsum(d[1 : p.max_period + 1, :], axis=-2) / tau
    den = cum_mean[p.min_period - 1 : p.max_period, :]
    return num / (den + TINY)


def parabolic_shifts(x):
    """_parabolic_interpolation along axis 0 (P5)."""
    shifts = np.zeros_like(x)
    xm, x0, xp = x[:-2], x[1:-1], x[2:]
    a = xp + xm - 2 * x0
    b = (xp - xm) / 2
    with np.errstate(divide="ignore", invalid="ignore"):
        s = -b / a
    s[np.abs(b) >= np.abs(a)] = 0
    shifts[1:-1] = s
    return shifts


def localmin0(x):
    """librosa.util.localmin along axis 0 for a 1-D vector."""
    out = np.zeros(x.shape, dtype=bool)
    out[1:-1] = (x[1:-1] < x[:-2]) & (x[1:-1] <= x[2:])
    out[-1] = x[-1] < x[-2]
    return out


# ----------------------------------------------------------------------------
# P6-P10: observation model
# ----------------------------------------------------------------------------
def observation(yin, shifts, p):
    """__pyin_helper: (obs float64[2*B, F], voiced_prob float64[F])."""
    n_lags, F = yin.shape
    B = p.n_pitch_bins
    yin_probs = np.zeros_like(yin)
    for i in range(F):
        col = yin[:, i]
        is_trough = localmin0(col)
        is_trough[0] = col[0] < col[1]
        (idx,) = np.nonzero(is_trough)
        if len(idx) == 0:
            continue
        h = col[idx]
        below = np.less.outer(h, p.thresholds[1:])
        pos = np.cumsum(below, axis=0) - 1
        n_tr = np.count_nonzero(below, axis=0)
        with np.errstate(all="ignore"):
            prior = scipy.stats.boltzmann.pmf(pos, p.boltzmann, n_tr)
        prior[~below] = 0
        # librosa: trough_prior.dot(beta_probs) -- a BLAS product whose summation order is
        # unspecified.  The restatement fixes it: products added in ascending threshold order.
        probs = np.zeros(len(idx))
        for j in range(p.n_thresholds):
            probs = probs + prior[:, j] * p.beta_probs[j]
        g = np.argmin(h)
        n_below_min = np.count_nonzero(~below[g, :])
        probs[g] += p.no_trough_prob * np.sum(p.beta_probs[:n_below_min])
        yin_probs[idx, i] = probs
    lag, fr = np.nonzero(yin_probs)
    period = p.min_period + lag
    period = period + shifts[lag, fr]
    f0c = p.sr / period
    bins = 12 * p.bins_per_semitone * np.log2(f0c / p.fmin)
    bins = np.clip(np.round(bins), 0, B).astype(int)
    obs = np.zeros((2 * B, F))
    obs[bins, fr] = yin_probs[lag, fr]
    voiced_prob = np.clip(np.sum(obs[:B, :], axis=0, keepdims=True), 0, 1)
    obs[B:, :] = (1 - voiced_prob) / B
    return obs, voiced_prob[0]


# ----------------------------------------------------------------------------
# P11: transition model and Viterbi
# ----------------------------------------------------------------------------
def transition_local_triangle(n_states, width):
    """sequence.transition_local(n_states, width, window='triangle', wrap=False)."""
    import scipy.signal

    T = np.zeros((n_states, n_states), dtype=np.float64)
    win = scipy.signal.get_window("triangle", width, fftbins=False)
    # util.pad_center(win, size=n_states)
    lpad = (n_states - width) // 2
    row = np.zeros(n_states)
    row[lpad : lpad + width] = win
    for i in range(n_states):
        r = np.roll(row, n_states // 2 + i + 1)
        lo, hi = max(0, i - width // 2), min(n_states, i + width // 2 + 1)
        r[:lo] = 0
        r[hi:] = 0
        T[i] = r
    T /= T.sum(axis=1, keepdims=True)
    return T


def transition_matrix(p):
    loc = transition_local_triangle(p.n_pitch_bins, p.transition_width)
    # sequence.transition_loop(2, 1 - switch_prob): off-diagonal = (1.0 - prob) / (n_states - 1)
    stay = 1 - p.switch_prob
    sw = np.array([[stay, (1.0 - stay) / 1], [(1.0 - stay) / 1, stay]])
    return np.kron(sw, loc)


_C = None


def _clib():
    global _C
    if _C is None:
        path = os.path.join(os.path.dirname(__file__), "_build", "liboracle.so")
        if os.path.exists(path):
            lib = ctypes.CDLL(path)
            lib.oracle_viterbi_dense.restype = None
            lib.oracle_viterbi_dense.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
            _C = lib
        else:
            _C = False
    return _C


def viterbi_states(log_prob, log_trans, log_p_init, use_c=True):
    """sequence._viterbi: log_prob [T, S]; first-max argmax; uint16 states."""
    T, S = log_prob.shape
    lib = _clib() if use_c else False
    if lib:
        lp = np.ascontiguousarray(log_prob, dtype=np.float64)
        lt = np.ascontiguousarray(log_trans, dtype=np.float64)
        li = np.ascontiguousarray(log_p_init, dtype=np.float64)
        st = np.zeros(T, dtype=np.int32)
        lib.oracle_viterbi_dense(lp.ctypes.data, lt.ctypes.data, li.ctypes.data, T, S, st.ctypes.data)
        return st.astype(np.uint16)
    state = np.zeros(T, dtype=np.uint16)
    ptr = np.zeros((T, S), dtype=np.uint16)
    value = log_prob[0] + log_p_init
    ltT = np.ascontiguousarray(log_trans.T)
    rows = np.arange(S)
    for t in range(1, T):
        trans_out = value + ltT          # [j, k] = value[k] + log_trans[k, j]
        am = np.argmax(trans_out, axis=1)
        ptr[t] = am
        value = log_prob[t] + trans_out[rows, am]
    state[-1] = np.argmax(value)
    for t in range(T - 2, -1, -1):
        state[t] = ptr[t + 1, state[t + 1]]
    return state


def initial_distribution(p, p_init="unvoiced"):
    """p_init handed to sequence.viterbi.  "unvoiced" (default) is librosa core/pitch.py::pyin as published
    (0.8 ... 0.10): `p_init = np.zeros(2 * n_pitch_bins); p_init[n_pitch_bins:] = 1 / n_pitch_bins` -- the chain
    starts in the unvoiced half, voiced states start at log(0 + tiny).  "uniform" is SURVEY.md P11's reading
    (1/(2B) everywhere), kept as the documented alternative.  Neither can be checked against an installed librosa
    here."""
    B = p.n_pitch_bins
    if p_init == "unvoiced":
        v = np.zeros(2 * B)
        v[B:] = 1 / B
        return v
    if p_init == "uniform":
        return np.ones(2 * B) / (2 * B)
    raise ValueError("p_init must be 'unvoiced' or 'uniform'")


def decode(obs, p, use_c=True, p_init="unvoiced"):
    """sequence.viterbi(obs, transition, p_init=p_init) -> states uint16[F]."""
    trans = transition_matrix(p)
    p_init = initial_distribution(p, p_init)
    log_trans = np.log(trans + TINY)
    log_prob = np.log(obs.T + TINY)
    log_p_init = np.log(p_init + TINY)
    return viterbi_states(log_prob, log_trans, log_p_init, use_c=use_c)


# ----------------------------------------------------------------------------
# the public call
# ----------------------------------------------------------------------------
def pyin(y, sr=44100, hop_length=512, fmin=82.4068892282175, fmax=1046.5022612023945,
         frame_length=2048, return_intermediates=False, use_c=True, block=2048, p_init="unvoiced"):
    """librosa.pyin(y, fmin=, fmax=, sr=, hop_length=) with every other argument
    at its 0.10 default.  Returns (f0 float64[F] with NaN unvoiced,
    voiced_flag bool[F], voiced_prob float64[F])."""
    y = np.asarray(y, dtype=np.float32)
    p = PyinParams(sr, fmin, fmax, frame_length, hop_length)
    y_frames = frame_centered(y, frame_length, hop_length)
    F = y_frames.shape[1]
    yin = np.empty((p.n_lags, F))
    for s in range(0, F, block):
        yin[:, s : s + block] = cmnd(y_frames[:, s : s + block], p)
    shifts = parabolic_shifts(yin)
    obs, voiced_prob = observation(yin, shifts, p)
    states = decode(obs, p, use_c=use_c, p_init=p_init)
    f0 = p.freqs[states % p.n_pitch_bins]
    voiced_flag = states < p.n_pitch_bins
    f0 = f0.copy()
    f0[~voiced_flag] = np.nan
    if return_intermediates:
        return f0, voiced_flag, voiced_prob, dict(yin=yin, shifts=shifts, obs=obs, states=states, params=p)
    return f0, voiced_flag, voiced_prob
