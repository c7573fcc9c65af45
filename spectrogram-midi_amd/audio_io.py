"""Decoded-PCM input for the engine: RIFF/WAVE via the stdlib (`librosa.load` is the reference's
loader, aegis_engine.py:24; mp3 and resampling are outside this path -- SURVEY.md 8f rank 4)."""
import wave

import numpy as np


def read_wav(path, sr, offset=0.0, duration=None):
    """-> float32 mono at `sr`.  int16/int32 PCM scale by 1/32768 and 1/2**31 (soundfile's
    convention, which librosa.load uses); 8-bit is unsigned; channels are averaged.  A file
    whose rate differs from `sr` is rejected rather than silently resampled differently."""
    with wave.open(path, "rb") as w:
        file_sr, ch, width, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        if file_sr != sr:
            raise ValueError(f"{path}: sample rate {file_sr} != engine rate {sr}; resampling "
                             "(librosa soxr_hq) is not part of the MI355X analyze path")
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
    return x


def write_wav(path, y, sr):
    """float32 [-1, 1) -> 16-bit PCM (test helper; mirrors soundfile.write's default subtype)."""
    pcm = np.clip(np.round(np.asarray(y, np.float64) * 32768.0), -32768, 32767).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes(pcm.tobytes())
