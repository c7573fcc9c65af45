"""Decoded-PCM input for the engine: RIFF/WAVE via the stdlib, `librosa.load` semantics (the reference's loader,
aegis_engine.py:24): offset/duration in native frames, mono = channel mean, then resampling
to the engine rate.  mp3 needs an external decoder and stays outside."""
import io
import math
import warnings
import wave

import numpy as np


def resample(y, orig_sr, target_sr):
    """librosa.resample(y, orig_sr, target_sr, res_type="polyphase", fix=True, scale=False): scipy's polyphase FIR,
    output trimmed / zero-padded to ceil(len * ratio) samples, float32.  librosa.load's default is "soxr_hq" (a
    different low-pass; libsoxr is not available here), so a resampled file differs from the reference's samples
    by the two filters' pass-band ripple (~1e-3) -- parity on this step is unpinned."""
    if orig_sr == target_sr:
        return np.asarray(y, np.float32)
    import scipy.signal
    g = math.gcd(int(orig_sr), int(target_sr))
    out = scipy.signal.resample_poly(np.asarray(y, np.float32), int(target_sr) // g, int(orig_sr) // g)
    n = int(np.ceil(len(y) * float(target_sr) / float(orig_sr)))
    out = out[:n] if len(out) >= n else np.pad(out, (0, n - len(out)))
    return np.ascontiguousarray(out, np.float32)


def read_wav_bytes(data, sr, offset=0.0, duration=None, resample_mismatch=True):
    """read_wav() for a WAV file held in memory (the auto-matcher's synthesised audio)."""
    return read_wav(io.BytesIO(data), sr, offset, duration, resample_mismatch)


def read_wav(path, sr, offset=0.0, duration=None, resample_mismatch=True):
    """-> float32 mono at `sr`.  int16/int32 PCM scale by 1/32768 and 1/2**31 (soundfile's
    convention, which librosa.load uses); 8-bit is unsigned; channels are averaged.  A file at another rate is
    resampled by resample() with a warning (or rejected when resample_mismatch=False)."""
    with wave.open(path, "rb") as w:
        file_sr, ch, width, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        if file_sr != sr and not resample_mismatch:
            raise ValueError(f"{path}: sample rate {file_sr} != engine rate {sr}")
        first = min(n, int(round(offset * file_sr)))
        count = n - first if duration is None else min(n - first, int(round(duration * file_sr)))
        w.setpos(first)
        raw = w.readframes(max(count, 0))
    if width == 2:
        x = np.frombuffer(raw, "<i2").astype(np.float32) / np.float32(32768.0)
    elif width == 4:
        x = (np.frombuffer(raw, "<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    elif width == 1:
        x = (np.frombuffer(raw, np.uint8).astype(np.float32) - 128.0) / np.float32(128.0)
    elif width == 3:
        b = np.frombuffer(raw, np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - 0x1000000, v)
        x = (v.astype(np.float64) / 8388608.0).astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported sample width {width}")
    if ch > 1:
        x = x.reshape(-1, ch).mean(axis=1).astype(np.float32)
    if file_sr != sr:
        warnings.warn(f"{path}: resampling {file_sr} -> {sr} Hz with a polyphase FIR (the reference uses soxr_hq)",
                      stacklevel=2)
        x = resample(x, file_sr, sr)
    return x


def write_wav(path, y, sr):
    """float32 [-1, 1) -> 16-bit PCM (test helper; mirrors soundfile.write's default subtype)."""
    pcm = np.clip(np.round(np.asarray(y, np.float64) * 32768.0), -32768, 32767).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes(pcm.tobytes())
