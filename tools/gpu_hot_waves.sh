#!/bin/bash
# hot scratch levels vs resident wavefronts (smaller hot set, less queueing): does the Infinity Cache serve them?
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r3_hot_waves.txt
: > $out
for waves in 2048 1024 512; do
  for hot in 0 2 3 4; do
    POLAR_HIP_HOT_LEVELS=$hot timeout -k 10 200 python3 tools/time_decode.py --M 16 --frames 262144 --fpw 128 --lds-level 6 --waves $waves --reps 2 2>&1 | grep "^M " | sed "s/^/hot $hot: /" | tee -a $out || exit 1
  done
done
