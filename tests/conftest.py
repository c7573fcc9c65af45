import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a clean checkout has no built artefacts (they are git-ignored): build them once (hipcc cross-compiles
    # gfx950 without a GPU; the oracle's C helper needs gcc only)
    lib = os.path.join(ROOT, "spectrogram-midi_amd", "libaegis_hip.so")
    helper = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    if not (os.path.exists(lib) and os.path.exists(helper)):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "spectrogram-midi_amd", "csrc")], check=True)
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


HAVE_GPU = _have_gpu()


def pytest_collection_modifyitems(config, items):
    if HAVE_GPU:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def gpu_handle():
    from spectrogram_midi_amd import _lib
    h = _lib.Handle(device=0)
    yield h
    h.close()


@pytest.fixture(scope="session")
def test_clips():
    """Small ragged batch covering the edge cases: fixture track, sweep, silence, a clip
    shorter than one frame, noise, an empty clip."""
    from tools import signals as S
    rng = np.random.default_rng(5)
    return {
        "guitar": S.guitar_test_track(),
        "sweep": S.sine_sweep(3.0),
        "silence": np.zeros(44100, np.float32),
        "tiny": (0.3 * np.sin(2 * np.pi * 220 * np.arange(700) / 44100)).astype(np.float32),
        "noise": rng.normal(0, 0.2, 30000).astype(np.float32),
        "empty": np.zeros(0, np.float32),
        "notes": S.guitar_clip(6.0, seed=11),
    }
