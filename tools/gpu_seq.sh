#!/bin/bash
# the same timing in consecutive processes on one fresh box: does the first process differ from the later ones?
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/r3_seq.txt
for i in 1 2 3 4 5; do
  timeout -k 10 300 python3 tools/time_decode.py --M 16 --frames 1048576 --reps 3 2>&1 | grep -E "^M " | sed "s/^/process $i: /" | tee -a gpurun_out/r3_seq.txt || exit 1
done
