"""Stage-by-stage parity of the HIP kernels (through the C ABI) against the CPU oracle on a
small ragged batch.  Bars: integer/boolean outputs exact; float64 pYIN intermediates to
~1e-9 (FFT rounding differs from pocketfft's); float32 RMS exact; mel power within 1e-4
relative to the clip maximum; dB image within 2e-3 dB."""
import numpy as np
import pytest

from oracle import dsp, engine as oengine, pyin as opyin, rake as orake

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def stage_handle():
    """A handle that also keeps the CMND rows (pyin_obs writes them only under AEGIS_DEBUG_STAGES=1)."""
    import os
    from spectrogram_midi_amd import _lib
    old = os.environ.get("AEGIS_DEBUG_STAGES")
    os.environ["AEGIS_DEBUG_STAGES"] = "1"
    try:
        h = _lib.Handle(device=0)
    finally:
        if old is None:
            del os.environ["AEGIS_DEBUG_STAGES"]
        else:
            os.environ["AEGIS_DEBUG_STAGES"] = old
    yield h
    h.close()


@pytest.fixture(scope="module")
def run(stage_handle, test_clips):
    gpu_handle = stage_handle
    names = list(test_clips)
    clips = [test_clips[k] for k in names]
    res = gpu_handle.analyze_batch(clips, rake_sensitivity=0.6)
    inter = {k: gpu_handle.debug_fetch(k) for k in ("dfn", "yin", "logobs", "logunv", "states", "melpow")}
    frames = [gpu_handle.frames_for(len(c)) for c in clips]
    # workspace rows follow the order the pass takes its clips in: longest first (stable); outputs keep the caller's order
    order = sorted(range(len(clips)), key=lambda i: (-frames[i], i))
    lo = np.zeros(len(clips), np.int64)
    pos = 0
    for i in order:
        lo[i] = pos
        pos += frames[i]
    offs = np.stack([lo, lo + np.array(frames)], axis=1)
    strides = {k: gpu_handle.param(k) for k in ("lag_stride", "yin_stride", "obs_stride")}
    ora = {}
    for k, c in zip(names, clips):
        p = opyin.PyinParams()
        yf = dsp.frame_centered(c, 2048, 512)
        acf, energy, d = opyin.difference_terms(yf, p)
        f0, vf, vp, it = opyin.pyin(c, return_intermediates=True)
        S = dsp.melspectrogram(c)
        ora[k] = dict(dfn=d[: p.max_period + 1], f0=f0, vf=vf, vp=vp, S=S, SdB=dsp.power_to_db(S),
                      rms=dsp.rms(c), **it)
    return dict(names=names, clips=clips, res=res, inter=inter, offs=offs, strides=strides, ora=ora)


def _rows(run, key, i, stride, width):
    a = run["inter"][key].reshape(-1, stride)
    return a[run["offs"][i][0]: run["offs"][i][1], :width]


def test_frame_counts(run):
    for k, r in zip(run["names"], run["res"]):
        assert len(r["f0"]) == len(run["ora"][k]["f0"]), k


def test_rms_bit_exact(run):
    for k, r in zip(run["names"], run["res"]):
        np.testing.assert_array_equal(r["rms"], run["ora"][k]["rms"], err_msg=k)


def test_difference_function(run):
    """d[tau] = (energy[0] + energy[tau]) - 2 acf[tau]: float32 running energies (bit-exact by construction) and the
    FFT autocorrelation (rounding differs from pocketfft's)."""
    for i, k in enumerate(run["names"]):
        got = _rows(run, "dfn", i, run["strides"]["lag_stride"], 537)
        ref = run["ora"][k]["dfn"].T
        scale = max(1.0, np.abs(ref).max())
        # |acf| < 1e-6 is clamped to 0 on both sides; a value within rounding of the clamp may fall either way
        err = np.abs(got - ref)
        assert err.max() <= 1e-9 * scale + 2.1e-6, (k, err.max())
        assert np.mean(err <= 1e-9 * scale) > 0.999, k


def test_cmnd(run):
    for i, k in enumerate(run["names"]):
        got = _rows(run, "yin", i, run["strides"]["yin_stride"], 495)
        ref = run["ora"][k]["yin"].T
        np.testing.assert_allclose(got, ref, rtol=1e-7, atol=1e-9, err_msg=k)


def test_observation(run):
    for i, k in enumerate(run["names"]):
        got = _rows(run, "logobs", i, run["strides"]["obs_stride"], 441)
        obs = run["ora"][k]["obs"]
        ref = np.log(obs[:441].T + opyin.TINY)
        assert np.array_equal(got > -700, ref > -700), (k, "observation support differs")
        np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-9, err_msg=k)
        lu = run["inter"]["logunv"][run["offs"][i][0]: run["offs"][i][1]]
        # 1 - voiced_prob cancels to rounding noise when every threshold finds a trough, so
        # the unvoiced observation is compared in the linear domain
        np.testing.assert_allclose(np.exp(lu), obs[441], rtol=1e-9, atol=1e-15, err_msg=k)
        np.testing.assert_allclose(run["res"][i]["voiced_prob"], run["ora"][k]["vp"], rtol=1e-10, atol=1e-12)
        # same scipy tables + same summation order: the prior masses are bit-identical, which is
        # what keeps log(1 - voiced_prob) from decaying into rounding noise (parity notes)
        np.testing.assert_array_equal(run["res"][i]["voiced_prob"], run["ora"][k]["vp"], err_msg=k)


def test_pitch_track_exact(run):
    for i, k in enumerate(run["names"]):
        r, o = run["res"][i], run["ora"][k]
        np.testing.assert_array_equal(r["voiced_flag"], o["vf"], err_msg=k)
        st = run["inter"]["states"][run["offs"][i][0]: run["offs"][i][1]]
        voiced = o["vf"]
        np.testing.assert_array_equal(st[voiced], o["states"][voiced], err_msg=k)   # pitch bins
        assert np.array_equal(np.isnan(r["f0"]), np.isnan(o["f0"])), k
        np.testing.assert_allclose(r["f0"][voiced], o["f0"][voiced], rtol=1e-14, err_msg=k)


def test_mel_and_db(run):
    for i, k in enumerate(run["names"]):
        got = run["inter"]["melpow"].reshape(-1, 128)[run["offs"][i][0]: run["offs"][i][1]].T
        ref = run["ora"][k]["S"]
        tol = 1e-4 * max(ref.max(), 1e-30)
        assert np.abs(got - ref).max() <= tol, (k, np.abs(got - ref).max(), ref.max())
        big = ref > 1e-6 * ref.max()
        if big.any():
            assert (np.abs(got - ref)[big] / ref[big]).max() <= 1e-4, k
        np.testing.assert_allclose(run["res"][i]["S_dB"], run["ora"][k]["SdB"], atol=2e-3, err_msg=k)


def test_rake_mask_exact(run):
    for i, k in enumerate(run["names"]):
        ref = orake.detect_rake_patterns(run["ora"][k]["SdB"], 512, 44100, 0.6)
        np.testing.assert_array_equal(run["res"][i]["rake_mask"], ref, err_msg=k)
        # and on the GPU's own dB image the reference rule gives the same mask
        ref2 = orake.detect_rake_patterns(run["res"][i]["S_dB"], 512, 44100, 0.6)
        np.testing.assert_array_equal(run["res"][i]["rake_mask"], ref2, err_msg=k)


def test_batch_equals_single(gpu_handle, test_clips, run):
    """Ragged batching must not change any clip's result."""
    for i, k in enumerate(run["names"]):
        single = gpu_handle.analyze_batch([test_clips[k]])[0]
        for key, v in single.items():
            np.testing.assert_array_equal(v, run["res"][i][key], err_msg=f"{k}/{key}")
