"""CPU baseline B of BASELINE.md: the reference's Turbo Mode (aegis_engine.py:183-216) with the oracle as the
per-chunk worker -- ProcessPoolExecutor(forkserver, max_workers=cores) over equal frame spans of one clip, plus
a clip-parallel variant (one whole clip per process, the best a CPU folder job can do).  One JSON line."""
import json, multiprocessing as mp, os, sys, time
from concurrent.futures import ProcessPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[k] = "1"
from oracle import engine as oe
from tools import signals


def whole_clip(seed):
    y = signals.guitar_clip(60.0, seed=seed)
    t0 = time.perf_counter()
    oe.audio_to_midi(y)
    return time.perf_counter() - t0


if __name__ == "__main__":
    cores = int(sys.argv[1]) if len(sys.argv) > 1 else os.cpu_count()
    ctx = mp.get_context("forkserver")
    y = signals.guitar_clip(180.0, seed=1)
    with ProcessPoolExecutor(max_workers=cores, mp_context=ctx) as pool:
        list(pool.map(abs, range(cores)))                       # start the workers (the reference pays this per call)
        t0 = time.perf_counter()
        oe.audio_to_midi(y, turbo_mode=True, num_cores=cores, pool=pool)
        t_turbo = time.perf_counter() - t0
        n = 2 * cores
        t0 = time.perf_counter()
        list(pool.map(whole_clip, range(n)))
        t_par = time.perf_counter() - t0
    print(json.dumps({"cores": cores, "turbo_one_180s_clip_audio_s_per_s": round(180.0 / t_turbo, 1),
                      "clip_parallel_audio_s_per_s": round(n * 60.0 / t_par, 1), "clips": n,
                      "note": "oracle (NumPy + C Viterbi) as the pyin worker; warm pool"}))
