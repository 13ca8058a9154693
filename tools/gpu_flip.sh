#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
o=gpurun_out/r3_relay_flip.txt
: > $o
for r in 0 1 0 1 0 1; do
  echo "flip $r" >> $o
  if [ $r = 1 ]; then export POLAR_HIP_RELAY_FLIP=1; else unset POLAR_HIP_RELAY_FLIP; fi
  timeout -k 10 300 python3 bench.py --steps 5 --warmup 1 --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['ber'])" >> $o || { cat $o; exit 1; }
done
cat $o
