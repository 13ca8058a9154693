#!/bin/bash
# the arena placement choice at work: kernel time per call in three consecutive processes, with and without it
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/r3_placement_tuner.txt
for i in 1 2 3; do
  for p in 1 0; do
    POLAR_HIP_PLACEMENT=$p POLAR_HIP_DEBUG=1 timeout -k 10 300 python3 tools/time_decode.py --M 16 --frames 1048576 --reps 8 2>&1 | grep -E "kernel|placement" | sed "s/^/process $i placement=$p: /" | tee -a gpurun_out/r3_placement_tuner.txt || exit 1
  done
done
