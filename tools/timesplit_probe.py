"""CPU experiment behind the time-split Viterbi: on the oracle's own observations,
(1) how many steps after a split point m does a run started L frames earlier from a guessed column agree with the true
column up to one additive constant (max - min of the difference <= sigma), and (2) how large are the decision margins
(best candidate - second best) along the decoded path -- the quantity that must exceed the accumulated rounding bound
for the speculative run's pointers to be provably those of the sequential run.

    python tools/timesplit_probe.py [seconds]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyin as op
from tools import signals


def forward(log_prob, ltT, v0, t0, t1, keep_margin_for=None):
    """columns of steps t0+1 .. t1 from column v0 at t0 -> (cols [t1 - t0, S], ptr)"""
    S = log_prob.shape[1]
    cols = np.empty((t1 - t0, S))
    ptr = np.empty((t1 - t0, S), np.int32)
    v = v0
    rows = np.arange(S)
    for t in range(t0 + 1, t1 + 1):
        c = v[None, :] + ltT
        am = np.argmax(c, axis=1)
        v = log_prob[t] + c[rows, am]
        cols[t - t0 - 1] = v
        ptr[t - t0 - 1] = am
    return cols, ptr


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
    L, SIGMA = 128, 1e-7
    for name, y in (("guitar", signals.guitar_clip(seconds, seed=1)),
                    ("polyphonic", signals.polyphonic_clip(seconds, seed=7)),
                    ("noisy", signals.guitar_clip(seconds, seed=3, noise_dbfs=-12.0)),
                    ("gaps", np.concatenate([signals.guitar_clip(seconds / 4, seed=5), np.zeros(int(44100 * seconds / 2), np.float32),
                                             signals.guitar_clip(seconds / 4, seed=6)]))):
        f0, vf, vp, im = op.pyin(y, return_intermediates=True)
        p, obs = im["params"], im["obs"]
        log_prob = np.log(obs.T + op.TINY)
        ltT = np.ascontiguousarray(np.log(op.transition_matrix(p) + op.TINY).T)
        lpi = np.log(op.initial_distribution(p) + op.TINY)
        T, S = log_prob.shape
        cols, ptr = forward(log_prob, ltT, log_prob[0] + lpi, 0, T - 1)
        cols = np.vstack([(log_prob[0] + lpi)[None], cols])          # cols[t] = column at frame t
        states = np.empty(T, np.int32)
        states[-1] = np.argmax(cols[-1])
        for t in range(T - 2, -1, -1):
            states[t] = ptr[t, states[t + 1]]                        # ptr[t] belongs to step t + 1
        assert np.array_equal(states, im["states"])
        # (2) margins along the path
        margins = np.empty(T - 1)
        for t in range(1, T):
            c = cols[t - 1] + ltT[states[t]]
            b = states[t - 1]
            best = c[b]
            c[b] = -np.inf
            margins[t - 1] = best - c.max()
        fin = np.sort(cols[-1])[::-1]
        q = np.quantile(margins, [0, 1e-4, 1e-3, 1e-2, 0.1, 0.5])
        print(f"{name}: {T} frames, voiced {vf.mean():.2f}; on-path margin min {q[0]:.3e} q1e-4 {q[1]:.3e} q1e-3 {q[2]:.3e} "
              f"q1e-2 {q[3]:.3e} q0.1 {q[4]:.3e} median {q[5]:.3f}; final arg-max margin {fin[0] - fin[1]:.3e}; "
              f"|V| max {np.abs(cols[-1][np.isfinite(cols[-1])]).max():.0f}", flush=True)
        # (1) lock-on after a split
        locks = []
        for m in range(512, T - 64, 512):
            w = m - L
            g, _ = forward(log_prob, ltT, log_prob[w] + lpi, w, min(T - 1, m + 1024))
            lock = None
            for t in range(m, min(T - 1, m + 1024) + 1, 16):
                d = cols[t] - g[t - w - 1]
                if d.max() - d.min() <= SIGMA:
                    lock = t - m
                    break
            locks.append(lock)
        print(f"   lock-on steps after the split (L = {L}, sigma = {SIGMA:g}): {locks}", flush=True)


if __name__ == "__main__":
    main()
