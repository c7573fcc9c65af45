"""Generates tests/golden/rake_golden.npz by running the REFERENCE's own
`aegis_engine_core/vision.py::detect_rake_patterns` (NumPy-only, importable in the
build container) on seeded synthetic dB images.  Run in the build container only
(`python tests/golden/make_rake_golden.py`); /root/reference does not travel."""
import importlib.util
import os

import numpy as np

REF = "/root/reference/aegis_engine_core/vision.py"


def main():
    spec = importlib.util.spec_from_file_location("ref_vision", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rng = np.random.default_rng(20260220)
    out = {}
    cases = []
    for i, (n_mels, F, sr, hop, ratio) in enumerate([
        (128, 360, 44100, 512, 0.6), (128, 862, 44100, 512, 0.5), (128, 400, 22050, 512, 0.6),
        (128, 300, 44100, 256, 0.6), (64, 257, 44100, 128, 0.4), (128, 1, 44100, 512, 0.6),
        (128, 2, 44100, 512, 0.6), (128, 50, 44100, 512, 0.6),
    ]):
        S = rng.uniform(-80, 0, size=(n_mels, F)).astype(np.float32)
        # sprinkle broadband bursts of assorted lengths, including one left open at the end
        t = 0
        while t < F:
            t += int(rng.integers(3, 25))
            L = int(rng.integers(1, 7))
            S[:, t:t + L] = rng.uniform(-15, 0, size=(n_mels, max(0, min(F, t + L) - t))).astype(np.float32)
            t += L
        if i in (0, 7):
            S[:, -2:] = -5.0          # open run at the end of the array -> dropped
        if i == 3:
            S[:, 40:48] = -70.0        # quiet columns: peak < -60 -> skipped
        mask = ref.detect_rake_patterns(S, hop, sr, ratio)
        out[f"S_{i}"] = S
        out[f"mask_{i}"] = np.asarray(mask, dtype=bool)
        cases.append((n_mels, F, sr, hop, ratio))
    out["cases"] = np.array(cases, dtype=np.float64)
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "rake_golden.npz"), **out)
    print("wrote rake_golden.npz with", len(cases), "cases")


if __name__ == "__main__":
    main()
