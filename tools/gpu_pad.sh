#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for pad in 0 1056 33824 0 1056 33824 263200; do
  POLAR_HIP_ARENA_PAD=$pad timeout -k 10 300 python3 tools/time_decode.py --M 16 --frames 1048576 --reps 3 2>&1 | grep "^M " | sed "s/^/pad $pad: /" | tee -a gpurun_out/r3_pad.txt || exit 1
done
