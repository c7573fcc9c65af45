"""Oracle: spectrogram-side arithmetic (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates what the reference asks librosa for in
`/root/reference/aegis_engine.py:25-26` (melspectrogram + power_to_db),
`aegis_engine.py:70` (feature.rms) and
`aegis_engine_core/midi_logic.py:51,69` (amplitude_to_db, hz_to_midi).
librosa is not vendored in the reference; the algorithms follow librosa 0.10.x
(`core/spectrum.py::stft/power_to_db/amplitude_to_db`, `filters.py::mel`,
`feature/spectral.py::melspectrogram/rms`, `core/convert.py`).  Parity unpinned.
"""
import numpy as np

# ----------------------------------------------------------------------------
# core/convert.py
# ----------------------------------------------------------------------------
_NOTE_PC = {"C": 0, "D": 2, "E": 4, "F": 5, "G": 7, "A": 9, "B": 11}


def note_to_midi(note):
    """librosa.note_to_midi for plain 'E2' / 'C#4' / 'Bb3' spellings."""
    pc = _NOTE_PC[note[0].upper()]
    i = 1
    while i < len(note) and note[i] in "#b":
        pc += 1 if note[i] == "#" else -1
        i += 1
    octave = int(note[i:]) if i < len(note) else 0
    return 12 * (octave + 1) + pc


def midi_to_hz(m):
    return 440.0 * (2.0 ** ((np.asanyarray(m, dtype=np.float64) - 69.0) / 12.0))


def note_to_hz(note):
    """`librosa.note_to_hz` (aegis_engine.py:63: 'E2' -> 82.4068892282175)."""
    return float(midi_to_hz(note_to_midi(note)))


def hz_to_midi(f):
    """`librosa.hz_to_midi`: 12*(log2(f) - log2(440)) + 69  (midi_logic.py:17,69)."""
    f = np.asanyarray(f)
    return 12 * (np.log2(f) - np.log2(440.0)) + 69


# ----------------------------------------------------------------------------
# filters.py
# ----------------------------------------------------------------------------
def hann_periodic(n):
    """scipy.signal.get_window('hann', n, fftbins=True) (general_cosine path)."""
    fac = np.linspace(-np.pi, np.pi, n + 1)
    w = np.zeros(n + 1)
    for k, a in enumerate((0.5, 0.5)):
        w += a * np.cos(k * fac)
    return w[:-1]


def _hz_to_mel_slaney(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if f.ndim:
        big = f >= min_log_hz
        mels[big] = min_log_mel + np.log(f[big] / min_log_hz) / logstep
    elif f >= min_log_hz:
        mels = min_log_mel + np.log(f / min_log_hz) / logstep
    return mels


def _mel_to_hz_slaney(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    big = m >= min_log_mel
    freqs[big] = min_log_hz * np.exp(logstep * (m[big] - min_log_mel))
    return freqs


def mel_filterbank(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    """librosa.filters.mel(htk=False, norm='slaney', dtype=float32)."""
    if fmax is None:
        fmax = float(sr) / 2
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float32)
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    mels = np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2)
    mel_f = _mel_to_hz_slaney(mels)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2 : n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


# ----------------------------------------------------------------------------
# core/spectrum.py + feature/spectral.py
# ----------------------------------------------------------------------------
def n_frames_for(n_samples, hop_length):
    """center=True framing: F = 1 + N // hop (SURVEY section 8 notation)."""
    return 1 + int(n_samples) // int(hop_length)


def frame_centered(y, frame_length, hop_length):
    """np.pad(..., mode='constant') by frame_length//2 then util.frame:
    a [frame_length, F] strided view (axis 0 is the unit-stride axis)."""
    y = np.asarray(y)
    ypad = np.pad(y, (frame_length // 2, frame_length // 2), mode="constant")
    n_frames = 1 + (len(ypad) - frame_length) // hop_length
    s = ypad.strides[0]
    return np.lib.stride_tricks.as_strided(
        ypad, shape=(frame_length, n_frames), strides=(s, hop_length * s), writeable=False
    )


def stft(y, n_fft=2048, hop_length=512, block=4096):
    """librosa.stft(center=True, pad_mode='constant', window='hann') -> complex64[1+n_fft/2, F].

    The window is float64, so window*frame and the rFFT are float64; the result
    is stored as complex64 (dtype_r2c of a float32 signal)."""
    y = np.asarray(y, dtype=np.float32)
    win = hann_periodic(n_fft).reshape(-1, 1)
    frames = frame_centered(y, n_fft, hop_length)
    out = np.zeros((1 + n_fft // 2, frames.shape[1]), dtype=np.complex64, order="F")
    for s in range(0, frames.shape[1], block):
        out[:, s : s + block] = np.fft.rfft(win * frames[:, s : s + block], axis=0)
    return out


def melspectrogram(y, sr=44100, n_fft=2048, hop_length=512, n_mels=128):
    """librosa.feature.melspectrogram(y, sr, n_fft, hop_length) -- power=2.0
    (aegis_engine.py:25).  float32 [n_mels, F]."""
    S = np.abs(stft(y, n_fft=n_fft, hop_length=hop_length)) ** 2.0
    mel_basis = mel_filterbank(sr, n_fft, n_mels=n_mels)
    return np.einsum("ft,mf->mt", S, mel_basis, optimize=True)


def power_to_db(S, amin=1e-10, top_db=80.0):
    """librosa.power_to_db(S, ref=np.max) (aegis_engine.py:26)."""
    S = np.asarray(S)
    ref_value = np.max(S)
    log_spec = 10.0 * np.log10(np.maximum(amin, S))
    log_spec -= 10.0 * np.log10(np.maximum(amin, ref_value))
    if top_db is not None:
        log_spec = np.maximum(log_spec, log_spec.max() - top_db)
    return log_spec


def amplitude_to_db(S, amin=1e-5, top_db=80.0):
    """librosa.amplitude_to_db(S, ref=np.max) (midi_logic.py:51):
    power_to_db(|S|**2, ref=max(|S|)**2, amin=amin**2)."""
    magnitude = np.abs(np.asarray(S))
    ref_value = np.max(magnitude)
    power = np.square(magnitude, out=magnitude.copy())
    amin2, ref2 = amin ** 2, ref_value ** 2
    log_spec = 10.0 * np.log10(np.maximum(amin2, power))
    log_spec -= 10.0 * np.log10(np.maximum(amin2, ref2))
    if top_db is not None:
        log_spec = np.maximum(log_spec, log_spec.max() - top_db)
    return log_spec


def rms(y, frame_length=2048, hop_length=512):
    """librosa.feature.rms(y=y, hop_length=hop)[0] (aegis_engine.py:70).
    float32[F]; the mean runs along the unit-stride axis of the framed view, so
    NumPy's pairwise float32 summation order applies."""
    y = np.asarray(y, dtype=np.float32)
    x = frame_centered(y, frame_length, hop_length)
    power = np.mean(np.square(x, dtype=np.float32), axis=-2, keepdims=True)
    return np.sqrt(power)[0]
