"""Standard MIDI File writer for the two-track (main / safe) output of
`AegisEngine.extract_events` (/root/reference/aegis_engine.py:98-179).  mido is not a
dependency: the chunk layout below is SMF 1.0 as mido writes it (type 1, 480 ticks per beat,
running status inside a track, end_of_track appended with delta 0)."""
import struct

import numpy as np

TICKS_PER_BEAT = 480
TICKS_PER_SECOND = 960.0   # second2tick(1.0, ticks_per_beat=480, tempo=500000)


class Track:
    def __init__(self):
        self._data = bytearray()
        self._status = None
        self._clock = 0

    @staticmethod
    def _varlen(value):
        if value < 0:
            raise ValueError("message time must be non-negative in MIDI file")
        groups = [value & 0x7F]
        value >>= 7
        while value:
            groups.append(0x80 | (value & 0x7F))
            value >>= 7
        return bytes(groups[::-1])

    def _emit(self, tick, status, *data):
        self._data += self._varlen(tick - self._clock)
        self._clock = tick
        if status != self._status:
            self._data.append(status)
            self._status = status
        self._data += bytes(d & 0x7F for d in data)

    def program_change(self, tick, program):
        self._emit(tick, 0xC0, program)

    def note_on(self, tick, note, velocity):
        self._emit(tick, 0x90, note, velocity)

    def note_off(self, tick, note, velocity=0):
        self._emit(tick, 0x80, note, velocity)

    def pitchwheel(self, tick, pitch):
        if not -8192 <= pitch <= 8191:
            raise ValueError("pitchwheel out of range")
        v = pitch + 8192
        self._emit(tick, 0xE0, v & 0x7F, v >> 7)

    def chunk(self):
        body = bytes(self._data) + b"\x00\xff\x2f\x00"
        return b"MTrk" + struct.pack(">I", len(body)) + body


def render(events, sr, hop_length, midi_program=27, vibrato_rate=5.0, vibrato_depth=0.3):
    """events -> SMF bytes.  Tick = int(frame * hop/sr * 960); hammer-on / pull-off scale the
    velocity by 0.6 / 0.5; bend = 15-point ease-out curve up to +-8191; vibrato = 10..20 sine
    points; all messages stable-sorted by tick, per-track delta times."""
    frame_ticks = hop_length / sr
    rows = []   # (tick, order, track, kind, a, b)
    for ev in events:
        t_on = int(ev["start"] * frame_ticks * TICKS_PER_SECOND)
        t_off = int(ev["end"] * frame_ticks * TICKS_PER_SECOND)
        technique, vel, trk = ev.get("technique"), ev["velocity"], ev["track"]
        if technique == "hammer_on":
            vel = int(vel * 0.6)
        elif technique == "pull_off":
            vel = int(vel * 0.5)
        rows.append((t_on, trk, "on", ev["note"], vel))
        rows.append((t_off, trk, "off", ev["note"], 0))
        length = t_off - t_on
        if technique == "bend":
            slope = ev.get("slope", 0.0)
            semis = min(2.0, abs(slope) * 10)
            top = int((1 if slope > 0 else -1) * (semis / 2.0) * 8191)
            for i in range(15):
                u = i / 15
                rows.append((t_on + int(u * length), trk, "pw", int(top * (1 - (1 - u) ** 2)), 0))
            rows.append((t_off, trk, "pw", 0, 0))
        elif technique == "vibrato":
            secs = length / TICKS_PER_SECOND
            count = max(10, min(20, int(secs * vibrato_rate * 4)))
            for i in range(count):
                angle = (i / count) * secs * vibrato_rate * 2 * np.pi
                rows.append((t_on + int((i / count) * length), trk, "pw",
                             int(np.sin(angle) * 8191 * vibrato_depth), 0))
            rows.append((t_off, trk, "pw", 0, 0))
    rows.sort(key=lambda r: r[0])

    tracks = {"main": Track(), "safe": Track()}
    for t in tracks.values():
        t.program_change(0, midi_program)
    for tick, trk, kind, a, b in rows:
        t = tracks["main" if trk == "main" else "safe"]
        if kind == "pw":
            t.pitchwheel(tick, a)
        elif kind == "on":
            t.note_on(tick, a, b)
        else:
            t.note_off(tick, a, b)
    header = b"MThd" + struct.pack(">IHHH", 6, 1, 2, TICKS_PER_BEAT)
    return header + tracks["main"].chunk() + tracks["safe"].chunk()
