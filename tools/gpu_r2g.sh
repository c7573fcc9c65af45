#!/bin/bash
# chunk-tail experiments
set -o pipefail
cd /root/repo
O=gpurun_out/r2g; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gputests.log; tail -4 $O/gputests.log
grep -q "pytest rc=0" $O/gputests.log || exit 1
run() { # name, args..., env via leading VAR=VAL
  n=$1; shift
  env "$@" timeout -k 10 400 python bench.py $BARGS --no-cpu-baseline > $O/bench_$n.log 2>&1 || { echo "$n failed"; tail -5 $O/bench_$n.log; return 1; }
  python - <<PY
import json
l=[x for x in open("$O/bench_$n.log") if x.startswith("{")][-1]; d=json.loads(l)
print("$n", d["ms_per_step"], d["value"], d["roofline"]["kernel_ms"])
PY
}
BARGS="--steps 5 --warmup 2"
run t256 AEGIS_X=0 && run t0 AEGIS_CHUNK_TAIL=0 && run t128 AEGIS_CHUNK_TAIL=128 && run t512 AEGIS_CHUNK_TAIL=512 && run t64 AEGIS_CHUNK_TAIL=64 && run k4096 AEGIS_TIME_CHUNK=4096 && run k1024 AEGIS_TIME_CHUNK=1024 || exit 1
BARGS="--steps 3 --warmup 1 --clips 256"
run c256_t256 AEGIS_X=0 && run c256_t0 AEGIS_CHUNK_TAIL=0 || exit 1
BARGS="--steps 2 --warmup 1 --config folder"
run folder_t256 AEGIS_X=0 && run folder_t0 AEGIS_CHUNK_TAIL=0
