"""Incremental analysis (aegis_stream_*, BASELINE configs[4]): whatever the push sizes, close() must return
arrays identical to the batch path on the whole signal, and the per-push rms / voiced_prob must already be the
final values.  The reference has no streaming path; parity is against our own offline result (itself checked
against the oracle elsewhere)."""
import numpy as np
import pytest

from spectrogram_midi_amd import _lib
from tools import signals

pytestmark = pytest.mark.gpu


def run_stream(h, y, sizes):
    st = h.open_stream(max_seconds=len(y) / h.sr + 1.0)
    parts, pos, i = [], 0, 0
    while pos < len(y):
        n = sizes[i % len(sizes)]
        parts.append(st.push(y[pos:pos + n]))
        pos += n
        i += 1
    final = st.close()
    st.free()
    return parts, final


@pytest.mark.parametrize("sizes", [[2048], [512], [1, 777, 4096, 30000], [100000]])
def test_stream_equals_offline(sizes):
    h = _lib.Handle()
    y = signals.guitar_clip(7.0, seed=31)
    ref = h.analyze_batch([y])[0]
    parts, final = run_stream(h, y, sizes)
    for k in ref:
        np.testing.assert_array_equal(final[k], ref[k], err_msg=f"{sizes} {k}")
    live_rms = np.concatenate([p["rms"] for p in parts])
    live_vp = np.concatenate([p["voiced_prob"] for p in parts])
    n = len(live_rms)
    assert 0 < n <= len(ref["rms"]) and len(ref["rms"]) - n <= 3        # the tail frames need the zero padding
    np.testing.assert_array_equal(live_rms, ref["rms"][:n])
    np.testing.assert_array_equal(live_vp, ref["voiced_prob"][:n])
    live = np.concatenate([p["live_state"] for p in parts])
    agree = np.mean((live < 441) == ref["voiced_flag"][:n])
    assert agree > 0.75                                                  # zero-lag decode: a preview, not the smoothed track (0.845 here)
    if sizes == [2048]:
        assert [len(p["rms"]) for p in parts[:3]] == [3, 4, 4]           # 2048-sample pushes = 4 hops
    h.close()


def test_stream_edge_cases():
    h = _lib.Handle()
    st = h.open_stream(max_seconds=1.0)
    assert len(st.push(np.zeros(100, np.float32))["rms"]) == 0
    out = st.close()
    assert len(out["f0"]) == 1 and not out["voiced_flag"].any()
    with pytest.raises(_lib.AegisError):
        st.push(np.zeros(10, np.float32))
    st.free()
    st = h.open_stream(max_seconds=0.1)
    with pytest.raises(_lib.AegisError):
        st.push(np.zeros(44100, np.float32))
    st.free()
    empty = h.open_stream(max_seconds=1.0)
    out = empty.close()
    np.testing.assert_array_equal(out["rms"], h.analyze_batch([np.zeros(0, np.float32)])[0]["rms"])
    empty.free()
    h.close()


def test_graph_push_after_plain_pushes():
    """A hop-multiple push (hipGraph replay) after odd-sized pushes (plain launches), and back: the device control
    block and the host counters must stay in step whichever path a push takes."""
    h = _lib.Handle()
    y = signals.guitar_clip(5.0, seed=32)
    ref = h.analyze_batch([y])[0]
    _, final = run_stream(h, y, [777, 2048, 2048, 1, 2048, 1535, 2048, 2048, 2048])
    for k in ref:
        np.testing.assert_array_equal(final[k], ref[k], err_msg=k)
    h.close()


def test_graph_knob_and_lifetime_orders(monkeypatch):
    """AEGIS_STREAM_GRAPH=0 keeps pushes on the plain-launch path (same results); a handle closed while a stream is
    still open stays alive until that stream is freed (no use-after-free in either order)."""
    y = signals.guitar_clip(3.0, seed=33)
    h = _lib.Handle()
    ref = h.analyze_batch([y])[0]
    monkeypatch.setenv("AEGIS_STREAM_GRAPH", "0")
    _, final = run_stream(h, y, [2048])
    monkeypatch.delenv("AEGIS_STREAM_GRAPH")
    for k in ref:
        np.testing.assert_array_equal(final[k], ref[k], err_msg=k)
    st = h.open_stream(max_seconds=1.0)
    st.push(y[:2048])
    h.close()                                   # aegis_destroy with a stream open: deferred
    with pytest.raises(_lib.AegisError):
        st.push(y[2048:4096])                   # the handle is gone as far as callers are concerned
    st.free()                                   # last stream: tears the handle down
    h2 = _lib.Handle()                          # the device is still usable
    assert len(h2.analyze_batch([y[:4096]])[0]["rms"]) == 9
    h2.close()
