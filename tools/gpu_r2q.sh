#!/bin/bash
set -o pipefail
cd /root/repo
O=gpurun_out/r2q; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_engine.py -m gpu -x -q -k "full_size or balanced" > $O/t1.log 2>&1; echo "t1 rc=$?"; tail -5 $O/t1.log
grep -q " passed" $O/t1.log || exit 1
run() {
  n=$1; shift
  env "$@" timeout -k 10 200 python bench.py $BARGS --no-cpu-baseline > $O/bench_$n.log 2>&1 || { echo "$n failed"; tail -5 $O/bench_$n.log; return 1; }
  python - <<PY
import json
l=[x for x in open("$O/bench_$n.log") if x.startswith("{")][-1]; d=json.loads(l)
print("$n", d["ms_per_step"], d["value"], d["roofline"]["kernel_ms"], d["roofline"]["launches_per_step"])
PY
}
BARGS="--steps 5 --warmup 2"
run pers AEGIS_X=0 && run nopers AEGIS_VITERBI_PERSISTENT=0 && run pers256 AEGIS_BALANCED_CHUNK=256 && run pers192 AEGIS_BALANCED_CHUNK=192 && run pers128 AEGIS_BALANCED_CHUNK=128
BARGS="--steps 5 --warmup 2"
run c64_1chunk AEGIS_BALANCED_CHUNK=0 AEGIS_TIME_CHUNK=65536
BARGS="--steps 3 --warmup 1 --clips 256"
run c256 AEGIS_X=0
