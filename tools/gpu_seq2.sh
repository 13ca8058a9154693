#!/bin/bash
# one long process, then short ones: is the speed-up of later processes a matter of time under load or of a new process?
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 400 python3 tools/time_decode.py --M 16 --frames 1048576 --reps 100 2>&1 | grep -E "kernel" | awk '{print $NF+0 > "/dev/null"; n++; if (n%10==1) print "long process call", n, $0}' | tee gpurun_out/r3_seq2.txt
for i in 2 3; do
  timeout -k 10 300 python3 tools/time_decode.py --M 16 --frames 1048576 --reps 3 2>&1 | grep -E "^M " | sed "s/^/process $i: /" | tee -a gpurun_out/r3_seq2.txt || exit 1
done
