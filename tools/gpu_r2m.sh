#!/bin/bash
cd /root/repo
export AEGIS_HIP_LIB=/root/repo/_ablate/lib_ab128.so AEGIS_BALANCED_CHUNK=0 AEGIS_TIME_CHUNK=65536
echo "== two workgroups per CU, single chunk"; timeout -k 10 300 python tools/frame_cycles.py 2>&1 | tail -9
echo "== one workgroup per CU"; AEGIS_FRAME_LDS_MIN=100000 timeout -k 10 300 python tools/frame_cycles.py 2>&1 | tail -9
