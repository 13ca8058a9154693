#!/bin/bash
# N = 2^12: all scratch levels hot and compact vs none, by resident wavefronts
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r3_hot_m12.txt
: > $out
for waves in 256 1024 2048; do
  for hot in 0 2 5; do
    POLAR_HIP_HOT_LEVELS=$hot timeout -k 10 300 python3 tools/time_decode.py --M 12 --frames 2097152 --fpw 128 --lds-level 6 --waves $waves --reps 2 2>&1 | grep "^M " | sed "s/^/hot $hot: /" | tee -a $out || exit 1
  done
done
