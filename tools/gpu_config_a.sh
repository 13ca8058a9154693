#!/bin/bash
# config A (N = 2^10, 2^22 frames): kernel shapes with and without hot levels, one box
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r3_config_A_shapes.txt
: > $out
for shape in "0 0 1" "128 6 1" "64 7 1" "64 7 4" "32 8 1" "32 8 4" "16 9 1" "256 5 1" "256 4 1"; do
  set -- $shape
  for hot in 0 9; do
    POLAR_HIP_HOT_LEVELS=$hot timeout -k 10 300 python3 tools/time_decode.py --M 10 --frames 4194304 --fpw $1 --lds-level $2 --team $3 --reps 3 2>&1 | grep "^M " | sed "s/^/hot $hot team $3: /" | tee -a $out || exit 1
  done
done
