"""Oracle: Standard MIDI File bytes (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates the SMF assembly in `/root/reference/aegis_engine.py:98-179` and the
byte encoding mido (un-vendored, un-pinned: requirements.txt:5) produces for it:
`MidiFile()` = type 1, 480 ticks/beat; per track delta-time VLQ + channel
message with running status, `end_of_track` meta appended with delta 0.
Parity unpinned (mido absent).
"""
import io
import struct

import numpy as np

TICKS_PER_BEAT = 480
TICKS_PER_SEC = 960.0          # mido.second2tick(1.0, 480, 500000)


def vlq(n):
    if n < 0:
        raise ValueError("message time must be non-negative in MIDI file")
    out = [n & 0x7F]
    n >>= 7
    while n:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    return bytes(reversed(out))


def _msg_bytes(m):
    kind = m[0]
    if kind == "program_change":
        return bytes([0xC0, m[1] & 0x7F])
    if kind == "note_on":
        return bytes([0x90, m[1] & 0x7F, m[2] & 0x7F])
    if kind == "note_off":
        return bytes([0x80, m[1] & 0x7F, m[2] & 0x7F])
    if kind == "pitchwheel":
        v = m[1] + 8192
        if not 0 <= v <= 16383:
            raise ValueError("pitchwheel value out of range")
        return bytes([0xE0, v & 0x7F, v >> 7])
    raise ValueError(kind)


def encode_track(msgs):
    """msgs: list of (delta_ticks, (kind, ...)).  mido.midifiles.write_track."""
    data, running = bytearray(), None
    for delta, m in msgs:
        data += vlq(int(delta))
        b = _msg_bytes(m)
        if b[0] == running:
            data += b[1:]
        else:
            data += b
        running = b[0]
    data += vlq(0) + b"\xff\x2f\x00"
    return b"MTrk" + struct.pack(">I", len(data)) + bytes(data)


def encode_file(tracks):
    head = b"MThd" + struct.pack(">IHHH", 6, 1, len(tracks), TICKS_PER_BEAT)
    return head + b"".join(encode_track(t) for t in tracks)


def timeline(events, sr, hop_length, vibrato_rate=5.0, vibrato_depth=0.3):
    """aegis_engine.py:105-160: flat list of (tick, kind, track, a, b) in stable tick order."""
    secs_per_frame = hop_length / sr
    rows = []
    for e in events:
        on = int(e["start"] * secs_per_frame * TICKS_PER_SEC)
        off = int(e["end"] * secs_per_frame * TICKS_PER_SEC)
        tech, vel, trk = e.get("technique"), e["velocity"], e["track"]
        if tech == "hammer_on":
            vel = int(vel * 0.6)
        elif tech == "pull_off":
            vel = int(vel * 0.5)
        rows.append((on, "note_on", trk, e["note"], vel))
        rows.append((off, "note_off", trk, e["note"], 0))
        span = off - on
        if tech == "bend":
            slope = e.get("slope", 0.0)
            depth = min(2.0, abs(slope) * 10)
            peak = int((1 if slope > 0 else -1) * (depth / 2.0) * 8191)
            for i in range(15):
                prog = i / 15
                rows.append((on + int(prog * span), "pitchwheel", trk, int(peak * (1 - (1 - prog) ** 2)), 0))
            rows.append((off, "pitchwheel", trk, 0, 0))
        elif tech == "vibrato":
            secs = span / TICKS_PER_SEC
            n = max(10, min(20, int(secs * vibrato_rate * 4)))
            for i in range(n):
                ph = (i / n) * secs * vibrato_rate * 2 * np.pi
                rows.append((on + int((i / n) * span), "pitchwheel", trk, int(np.sin(ph) * 8191 * vibrato_depth), 0))
            rows.append((off, "pitchwheel", trk, 0, 0))
    rows.sort(key=lambda r: r[0])
    return rows


def write_smf(events, sr, hop_length, midi_program=27, vibrato_rate=5.0, vibrato_depth=0.3):
    """-> bytes of the 2-track (main, safe) file aegis_engine.py:98-179 saves."""
    tracks = {"main": [(0, ("program_change", midi_program))], "safe": [(0, ("program_change", midi_program))]}
    last = {"main": 0, "safe": 0}
    for tick, kind, trk, a, b in timeline(events, sr, hop_length, vibrato_rate, vibrato_depth):
        key = "main" if trk == "main" else "safe"
        msg = ("pitchwheel", a) if kind == "pitchwheel" else (kind, a, b)
        tracks[key].append((tick - last[key], msg))
        last[key] = tick
    return encode_file([tracks["main"], tracks["safe"]])


def parse_smf(blob):
    """Minimal reader used by tests: -> (type, tpb, [[(delta, status, data bytes)...]...])."""
    f = io.BytesIO(blob)
    assert f.read(4) == b"MThd"
    _, typ, ntr, tpb = struct.unpack(">IHHH", f.read(10))
    tracks = []
    for _ in range(ntr):
        assert f.read(4) == b"MTrk"
        (n,) = struct.unpack(">I", f.read(4))
        d = f.read(n)
        i, running, msgs = 0, None, []
        while i < len(d):
            delta = 0
            while True:
                c = d[i]; i += 1
                delta = (delta << 7) | (c & 0x7F)
                if not c & 0x80:
                    break
            if d[i] == 0xFF:
                typ_, ln = d[i + 1], d[i + 2]
                msgs.append((delta, 0xFF, bytes(d[i + 1 : i + 3 + ln])))
                i += 3 + ln
                running = None
                continue
            if d[i] & 0x80:
                running = d[i]; i += 1
            nbytes = 1 if (running & 0xF0) in (0xC0, 0xD0) else 2
            msgs.append((delta, running, bytes(d[i : i + nbytes])))
            i += nbytes
        tracks.append(msgs)
    return typ, tpb, tracks
