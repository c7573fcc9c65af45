"""Oracle rake mask vs golden masks produced by the reference's own vision.py
(tests/golden/make_rake_golden.py).  CPU only."""
import os

import numpy as np

from oracle import rake

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "rake_golden.npz"))


def test_oracle_matches_reference_goldens():
    for i, (n_mels, F, sr, hop, ratio) in enumerate(G["cases"]):
        got = rake.detect_rake_patterns(G[f"S_{i}"], int(hop), int(sr), float(ratio))
        np.testing.assert_array_equal(got, G[f"mask_{i}"], err_msg=f"case {i}")


def test_run_length_quirks():
    # vision.py:27-36: open run at the end dropped; 44.1k/512 -> min 0, max 2 frames
    assert rake.run_length_window(512, 44100) == (0, 2)
    assert rake.run_length_window(512, 22050) == (0, 1)
    f = np.array([1, 1, 0, 1, 1, 1, 0, 1, 0, 0, 1, 1], bool)
    np.testing.assert_array_equal(rake.keep_short_runs(f, 0, 2),
                                  np.array([1, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0], bool))
