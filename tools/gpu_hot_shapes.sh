#!/bin/bash
# hot levels for other kernel shapes (single-wavefront groups), 2^20 frames of N = 2^16, one box
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r3_hot_shapes.txt
: > $out
for shape in "64 7" "32 8" "128 5" "256 5"; do
  set -- $shape
  for hot in 0 3 4 5; do
    POLAR_HIP_HOT_LEVELS=$hot timeout -k 10 300 python3 tools/time_decode.py --M 16 --frames 1048576 --fpw $1 --lds-level $2 --reps 2 2>&1 | grep "^M " | sed "s/^/hot $hot: /" | tee -a $out || exit 1
  done
done
