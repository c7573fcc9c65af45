"""CPU checks of the C-ABI library: it loads, exports every symbol include/aegis_hip.h declares,
builds its host tables without touching a GPU (device=-1), those tables match the NumPy/SciPy
constructions of the oracle, and analyze calls fail loudly without a device."""
import ctypes
import os
import re

import numpy as np
import pytest
import scipy.stats

from oracle import dsp, pyin as opyin
from spectrogram_midi_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_handle():
    h = _lib.Handle(device=-1, scipy_tables=False)      # the library's built-in closed-form tables
    yield h
    h.close()


def test_python_binding_feeds_scipy_tables():
    """The binding replaces the prior tables with the arrays librosa itself would build, bit for bit."""
    h = _lib.Handle(device=-1)
    p = opyin.PyinParams()
    np.testing.assert_array_equal(h.table("beta_probs"), p.beta_probs)
    np.testing.assert_array_equal(h.table("beta_cumsum"), [np.sum(p.beta_probs[:n]) for n in range(101)])
    np.testing.assert_array_equal(h.table("freqs"), p.freqs)
    fact, ex = h.table("boltz_fact"), h.table("boltz_exp")
    for N in (1, 2, 3, 17, 100, 248):
        np.testing.assert_array_equal(fact[N] * ex[:N], scipy.stats.boltzmann.pmf(np.arange(N), 2.0, N))
    with pytest.raises(_lib.AegisError):
        h.set_table("beta_probs", np.zeros(7))
    with pytest.raises(_lib.AegisError):
        h.set_table("hann", np.zeros(2048))
    h.close()


def test_header_symbols_are_exported():
    hdr = open(os.path.join(ROOT, "include", "aegis_hip.h")).read()
    declared = set(re.findall(r"\b(aegis_[a-z_]+)\s*\(", hdr))
    lib = _lib.load()
    assert declared and declared == set(_lib.EXPORTS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.aegis_abi_version() == 2


def test_geometry(host_handle):
    h = host_handle
    assert [h.param(k) for k in ("min_period", "max_period", "n_lags", "n_pitch_bins", "transition_width")] == \
        [42, 536, 495, 441, 51]
    assert h.frames_for(0) == 1 and h.frames_for(184014) == 360 and h.frames_for(7938000) == 15504


def test_tables_match_numpy_scipy(host_handle):
    h = host_handle
    p = opyin.PyinParams()
    np.testing.assert_array_equal(h.table("hann"), dsp.hann_periodic(2048))
    np.testing.assert_array_equal(h.table("mel_dense").reshape(128, 1025), dsp.mel_filterbank(44100, 2048))
    np.testing.assert_array_equal(h.table("thresholds"), p.thresholds)
    np.testing.assert_allclose(h.table("beta_probs"), p.beta_probs, rtol=1e-12)
    np.testing.assert_allclose(h.table("beta_cumsum"), [np.sum(p.beta_probs[:n]) for n in range(101)], rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(h.table("beta_suffix"), [np.sum(p.beta_probs[n:]) for n in range(101)], rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(h.table("freqs"), p.freqs, rtol=4e-16)
    fact, ex = h.table("boltz_fact"), h.table("boltz_exp")
    for N in (1, 2, 5, 31, 248):
        np.testing.assert_allclose(fact[N] * ex[:N], scipy.stats.boltzmann.pmf(np.arange(N), 2.0, N), rtol=1e-15)
    tw = h.table("twiddle").reshape(-1, 2)
    np.testing.assert_allclose(tw[:, 0] + 1j * tw[:, 1], np.exp(-2j * np.pi * np.arange(2048) / 2048), atol=1e-15)


def test_banded_transition_table(host_handle):
    """The kernels use a 51-class band-compressed log-transition table: edge rows exact, interior
    rows share one representative normalisation (<= 1 ulp from librosa's per-row sums)."""
    p = opyin.PyinParams()
    LT = np.log(opyin.transition_matrix(p) + opyin.TINY)
    B, H, W = 441, 25, 51
    band = host_handle.table("log_trans_band").reshape(4, 51, 51)
    cls = lambda b: b if b < H else (b - (B - 1 - 2 * H) if b > B - 1 - H else H)
    worst = 0.0
    for v in range(2):
        for v2 in range(2):
            for b in range(B):
                js = np.arange(max(0, b - H), min(B, b + H + 1))
                ref = LT[v * B + b, v2 * B + js]
                got = band[v * 2 + v2, cls(b), js - b + H]
                worst = max(worst, np.abs(ref - got).max())
                if b < H or b > B - 1 - H:
                    np.testing.assert_array_equal(got, ref)
    assert worst < 4e-15
    # everything outside the band is log(tiny)
    assert LT[0, 200] == np.log(opyin.TINY)


@pytest.mark.parametrize("sr,H", [(44100, 25), (22050, 50)])
def test_packed_transition_table(sr, H):
    """The LDS copy of the band table (csrc/viterbi.hip pk_*): blocks (0,0) == (1,1) and (0,1) == (1,0) are stored once,
    edge rows only over the targets that exist.  Every entry of the full table must be found at its packed index."""
    h = _lib.Handle(sample_rate=sr, device=-1)
    W, B = 2 * H + 1, 441
    assert h.param("transition_width") == W
    full = h.table("log_trans_band").reshape(4, W, W)
    np.testing.assert_array_equal(full[0], full[3])
    np.testing.assert_array_equal(full[1], full[2])
    NP = 3 * H * H + 3 * H + 2
    pack = h.table("log_trans_pack").reshape(2, NP)
    lo_start = lambda e: 1 + e * (H + 1) + e * (e - 1) // 2
    int_start = 1 + H * (H + 1) + H * (H - 1) // 2
    hi_start = lambda e: int_start + W + 2 * H * e - e * (e - 1) // 2
    assert hi_start(H) == NP
    for q in range(2):
        for e in range(H):                                   # source bin e: targets 0 .. e+H  <=>  dd = H-e .. 2H
            np.testing.assert_array_equal(pack[q, lo_start(e):lo_start(e) + H + 1 + e], full[q, e, H - e:])
            # source bin B-H+e (class H+1+e): targets .. B-1  <=>  dd = 0 .. 2H-1-e
            np.testing.assert_array_equal(pack[q, hi_start(e):hi_start(e) + 2 * H - e], full[q, H + 1 + e, :2 * H - e])
        np.testing.assert_array_equal(pack[q, int_start:int_start + W], full[q, H])
    h.close()


def test_other_configuration_and_rejections():
    h = _lib.Handle(sample_rate=22050, device=-1)
    assert (h.param("min_period"), h.param("max_period"), h.param("transition_width")) == (21, 268, 101)
    np.testing.assert_array_equal(h.table("mel_dense").reshape(128, 1025), dsp.mel_filterbank(22050, 2048))
    h.close()
    with pytest.raises(_lib.AegisError):
        _lib.Handle(n_fft=1024, device=-1)
    with pytest.raises(_lib.AegisError):
        _lib.Handle(fmin=500.0, fmax=100.0, device=-1)


def test_no_cpu_fallback(host_handle):
    with pytest.raises(_lib.AegisError) as e:
        host_handle.analyze_batch([np.zeros(1000, np.float32)])
    assert e.value.code == _lib.ERR_DEVICE


def test_pyin_init_is_a_create_time_choice():
    """aegis_config.pyin_init: 0 (default) = librosa's unvoiced start, 1 = uniform; anything else is refused."""
    for mode, code in (("unvoiced", 0), ("uniform", 1)):
        h = _lib.Handle(device=-1, pyin_init=mode)
        assert h.param("pyin_init") == code and h.pyin_init == mode
        h.close()
    h = _lib.Handle(device=-1)
    assert h.param("pyin_init") == 0
    h.close()
    cfg = _lib.Config(44100, 512, 2048, 128, 0.0, 0.0, -1, 7, 0)
    out = ctypes.c_void_p()
    assert _lib.load().aegis_create(ctypes.byref(cfg), ctypes.byref(out)) == _lib.ERR_INVALID
    assert b"pyin_init" in _lib.load().aegis_last_error(None)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "spectrogram-midi_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), fn


def test_committed_bench_line_has_the_contract_fields():
    """profiles/r2_bench.json is bench.py's own output on the MI355X: metric / value / n_gpus / steps / warmup /
    ms_per_step / scaling / dtype / data / config.workload plus the roofline and cpu_baseline objects."""
    import json, os
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r2_bench.json")
    d = json.load(open(p))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5 and r["unit"] == "GB/s"
    assert r["traffic"] is None or r["traffic"] > 0
    assert "traffic_static" in r and 0 < r["cus_busy_fraction"] <= 1
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert c["turbo"]["cores"] >= 1 and c["turbo"]["value"] > 0           # the reference's Turbo Mode beside the stable path
    assert abs(d["value"] - 64 * 180 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    assert len(d["rank_busy_ms"]) == d["n_gpus"] and "events" in d


def _build_c_demo(tmp_path):
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "aegis_demo")
    libdir = os.path.join(root, "spectrogram-midi_amd")
    subprocess.run(["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", "aegis_demo.c"), "-o", exe, "-L" + libdir, "-l:libaegis_hip.so",
                    "-Wl,-rpath," + libdir, "-lm"], check=True, capture_output=True)
    return exe


def test_c_abi_example_compiles_and_links(tmp_path):
    """The boundary is a C ABI: examples/aegis_demo.c (plain C99, no Python, no torch) must build against
    include/aegis_hip.h and link against the shared library."""
    import os
    assert os.path.exists(_build_c_demo(tmp_path))


@pytest.mark.gpu
def test_c_abi_example_runs(tmp_path):
    import subprocess
    r = subprocess.run([_build_c_demo(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "on the 220.00 Hz grid point" in r.stdout


def test_no_exception_crosses_the_c_boundary(host_handle):
    """include/aegis_hip.h: nothing is thrown across the boundary.  Every exported body runs inside a
    try/catch that maps the exception to a code; the debug entry's test hooks throw on purpose."""
    import ctypes as C
    lib, h = host_handle.lib, host_handle._h
    for hook, code, text in ((b"throw_bad_alloc", _lib.ERR_NOMEM, "out of host memory"),
                             (b"throw_length_error", _lib.ERR_NOMEM, "too large"),
                             (b"throw_runtime_error", _lib.ERR_DEVICE, "test hook: runtime_error"),
                             (b"throw_int", _lib.ERR_DEVICE, "unknown C++ exception")):
        assert lib.aegis_debug_fetch(h, hook, None, 0) == code
        assert text in lib.aegis_last_error(h).decode()
    # the handle is still usable afterwards
    assert host_handle.param("n_pitch_bins") == 441
    # a table request the library cannot size reports a code instead of unwinding
    cfg = _lib.Config(44100, 512, 2048, 1 << 30, 0.0, 0.0, -1, 0, 0)
    out = C.c_void_p()
    assert lib.aegis_create(C.byref(cfg), C.byref(out)) == _lib.ERR_INVALID and not out.value


def test_pass_throughs_forward_to_the_reference_package(tmp_path, monkeypatch):
    """aegis_engine.py:29-36 forward to aegis_engine_core.stems / tabs: with that package importable (the drop-in
    case) the calls are forwarded unchanged, without it they raise NotImplementedError."""
    import importlib
    import sys
    from spectrogram_midi_amd.engine import AegisEngine
    eng = AegisEngine()
    for mod in [m for m in sys.modules if m.split(".")[0] == "aegis_engine_core"]:
        sys.modules.pop(mod)
    with pytest.raises(NotImplementedError):
        eng.generate_tabs([])
    pkg = tmp_path / "aegis_engine_core"
    pkg.mkdir()
    (pkg / "__init__.py").write_text("")
    (pkg / "tabs.py").write_text("def generate_tabs(events):\n    return ('tabs', len(events))\n"
                                 "def export_musicxml(tab_data, path):\n    return ('xml', tab_data, path)\n")
    (pkg / "stems.py").write_text("def separate_stems(wav, out):\n    return ('stems', wav, out)\n")
    monkeypatch.syspath_prepend(str(tmp_path))
    importlib.invalidate_caches()
    assert eng.generate_tabs([1, 2, 3]) == ("tabs", 3)
    assert eng.export_musicxml("t", "x.xml") == ("xml", "t", "x.xml")
    assert eng.separate_stems("a.wav", "out") == ("stems", "a.wav", "out")
    for mod in [m for m in sys.modules if m.split(".")[0] == "aegis_engine_core"]:
        sys.modules.pop(mod)          # not monkeypatch.delitem: its teardown would put the stand-ins back


def test_representative_interior_rows_decode_like_the_true_matrix(host_handle):
    """33 of the 391 interior rows of librosa's transition matrix are normalised by a row sum one ulp away from the one
    the kernels' band table uses (1 056 of 79 764 in-band entries differ, by < 4e-15 in the log).  The decoded path is what
    leaves the Viterbi: the oracle's dense decoder run with the TRUE matrix and with the matrix rebuilt from the kernels'
    table must give the same states on pYIN observations of tonal, polyphonic and noisy clips and on random sparse
    observation sequences (where near-ties are likeliest)."""
    from tools import signals
    p = opyin.PyinParams()
    B, H, S = 441, 25, 882
    LT_true = np.log(opyin.transition_matrix(p) + opyin.TINY)
    band = host_handle.table("log_trans_band").reshape(4, 51, 51)
    cls = lambda b: b if b < H else (b - (B - 1 - 2 * H) if b > B - 1 - H else H)
    LT_rep = np.full((S, S), np.log(opyin.TINY))
    for v in range(2):
        for v2 in range(2):
            for b in range(B):
                js = np.arange(max(0, b - H), min(B, b + H + 1))
                LT_rep[v * B + b, v2 * B + js] = band[v * 2 + v2, cls(b), js - b + H]
    differing = LT_rep != LT_true
    assert 0 < differing.sum() <= 4 * 1056 and np.abs(LT_rep - LT_true).max() < 4e-15
    log_p_init = np.log(np.ones(S) / S + opyin.TINY)
    seqs = []
    rng = np.random.default_rng(11)
    noisy = signals.guitar_clip(6.0, seed=3) + (10 ** (-12 / 20) * rng.standard_normal(6 * 44100)).astype(np.float32)
    for y in (signals.guitar_clip(8.0, seed=2), signals.polyphonic_clip(6.0, seed=103), noisy):
        obs = opyin.pyin(y, return_intermediates=True)[3]["obs"]
        seqs.append(np.log(obs.T + opyin.TINY))
    for k in range(3):                                       # random sparse rows: a few observed bins, arbitrary masses
        T = 400
        obs = np.zeros((T, S))
        for t in range(T):
            n = int(rng.integers(0, 12))
            bins = rng.integers(0, B, n)
            w = rng.random(n) * rng.random()
            obs[t, bins] = w / max(w.sum(), 1e-12) * rng.random()
            obs[t, B:] = max(0.0, 1.0 - obs[t, :B].sum()) / B
        seqs.append(np.log(obs + opyin.TINY))
    for lp in seqs:
        a = opyin.viterbi_states(lp, LT_true, log_p_init)
        b = opyin.viterbi_states(lp, LT_rep, log_p_init)
        np.testing.assert_array_equal(a, b)


def test_graft_entry_imports_resolve():
    """Everything __graft_entry__.smoke() imports exists where it says (the GPU suite runs smoke() itself)."""
    import ast
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.ImportFrom) and node.module and node.level == 0:
            mod = __import__(node.module, fromlist=[a.name for a in node.names])
            for a in node.names:
                assert hasattr(mod, a.name), (node.module, a.name)
