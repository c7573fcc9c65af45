export AEGIS_TRACE_DESTROY=1
run() { echo "=== EXPERIMENT=$AEGIS_DESTROY_EXPERIMENT $*"; timeout -k 5 25 python tools/exit_hang_probe.py "$@" 2>&1 | grep -v "^Extension modules" | tail -4; echo "rc=${PIPESTATUS[0]}"; }
run close 44100 1 180 1 0 device
AEGIS_CU_SPLIT=0 run close 44100 1 180 1 0 host
export AEGIS_DESTROY_EXPERIMENT=1; run close 44100 1 180 1 0 host
export AEGIS_DESTROY_EXPERIMENT=2; run close 44100 1 180 1 0 host
export AEGIS_DESTROY_EXPERIMENT=3; run close 44100 1 180 1 0 host
export AEGIS_DESTROY_EXPERIMENT=4; run close 44100 1 180 1 0 host
