#!/bin/bash
# relay launches decided by the clock: what the handle sees and keeps, for several workloads
export TMPDIR=/tmp
mkdir -p gpurun_out
o=gpurun_out/r3_relay_by_the_clock.txt
: > $o
for args in "" "" "--frames 786432" "--frames 524288" "--config A" "--config D"; do
  echo "bench.py $args" >> $o
  POLAR_HIP_DEBUG=1 timeout -k 10 400 python3 bench.py $args --steps 5 --warmup 1 --cpu-seconds 0 2> gpurun_out/tuned.err | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('launches_per_decode'), d['ber'])" >> $o || { cat $o; tail gpurun_out/tuned.err; exit 1; }
  grep -E "arena placement|relay launches [0-9]|relay stream" gpurun_out/tuned.err >> $o
done
cat $o
