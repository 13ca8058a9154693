#!/bin/bash
# Sweep of the number of scratch levels that keep the default cache policy (POLAR_HIP_HOT_LEVELS)
# for the three wide kernel shapes, one box, N = 2^16, 2^19 frames.
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r3_hot_sweep.txt
: > $out
[ -n "$SKIP_PARITY" ] || timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -k "digest or kat or grid" > gpurun_out/r3_hot_parity.log 2>&1 || { tail -30 gpurun_out/r3_hot_parity.log; exit 1; }
tail -2 gpurun_out/r3_hot_parity.log
for shape in ${SHAPES:-"128 6 1" "64 7 4" "32 8 4"}; do
  set -- $shape
  for hot in ${HOTS:-0 1 2 3 4 5 6}; do
    POLAR_HIP_HOT_LEVELS=$hot timeout -k 10 200 python3 tools/time_decode.py --M 16 --frames 524288 --fpw $1 --lds-level $2 --team $3 --reps 3 2>&1 | grep "^M " | sed "s/^/hot $hot team $3: /" | tee -a $out || exit 1
  done
done
