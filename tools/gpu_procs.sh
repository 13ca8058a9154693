#!/bin/bash
# the headline bench in consecutive processes: kernel time and what the placement choice saw
export TMPDIR=/tmp
mkdir -p gpurun_out
o=gpurun_out/r3_procs.txt
: > $o
for i in 1 2 3 4 5; do
  POLAR_HIP_DEBUG=1 timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 $* 2> gpurun_out/procs.err | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('process $i:', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" >> $o || { tail gpurun_out/procs.err; exit 1; }
  grep "arena placement" gpurun_out/procs.err >> $o
done
cat $o
