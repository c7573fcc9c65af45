#!/bin/bash
# One GPU session after a kernel or schedule change: the whole -m gpu suite, then the bench lines that matter
# (headline, the same workload as a single time chunk = every kernel alone on the chip, 256 clips, the folder).
#   gpurun --timeout 1200 -- 'bash tools/gpu_check.sh [VAR=VALUE ...]'      (the variables apply to every bench run)
set -o pipefail
cd /root/repo
O=gpurun_out/check; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gputests.log; tail -3 $O/gputests.log
grep -q "pytest rc=0" $O/gputests.log || exit 1
run() {
  n=$1; shift
  env "$@" "${EXTRA_ENV[@]}" timeout -k 10 400 python bench.py $BARGS --no-cpu-baseline --no-extras > $O/bench_$n.log 2>&1 || { echo "$n failed"; tail -5 $O/bench_$n.log; return 1; }
  python - <<PY
import json
l=[x for x in open("$O/bench_$n.log") if x.startswith("{")][-1]; d=json.loads(l)
print("$n", d["ms_per_step"], d["value"], d["roofline"]["kernel_ms"], d.get("viterbi_list_only_rate"))
if d.get("persistent_fallbacks", 0):      # the single Viterbi launch gave up waiting: the line above is the slower schedule
    print("WARNING: $n persistent_fallbacks =", d["persistent_fallbacks"]); raise SystemExit(3)
PY
}
EXTRA_ENV=("$@"); [ ${#EXTRA_ENV[@]} -eq 0 ] && EXTRA_ENV=(AEGIS_X=0)
BARGS="--config shard --steps 8 --warmup 2"
run c64 AEGIS_X=0; run c64_1chunk AEGIS_BALANCED_CHUNK=0 AEGIS_TIME_CHUNK=65536
BARGS="--config shard --steps 3 --warmup 1 --clips 256"
run c256 AEGIS_X=0
BARGS="--steps 2 --warmup 1 --config folder"
run folder AEGIS_X=0
