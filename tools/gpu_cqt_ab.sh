#!/bin/bash
# CQT kernel time per queue-depth build: bash tools/gpu_cqt_ab.sh <lib suffixes...>   (prod = the product library)
cd /root/repo
for a in "$@"; do
  lib=/root/repo/_ablate/lib_ab$a.so; [ "$a" = "prod" ] && lib=/root/repo/spectrogram-midi_amd/libaegis_hip.so
  echo -n "$a "; AEGIS_HIP_LIB=$lib timeout -k 10 200 python tools/bench_cqt.py 2>&1 | tail -1
done
