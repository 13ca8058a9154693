#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
o=gpurun_out/r3_relay.txt
: > $o
for r in 1 2 4; do
  echo "POLAR_HIP_RELAY=$r" >> $o
  POLAR_HIP_RELAY=$r timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'] if 'kernel_ms' in d['roofline'] else '')" >> $o || exit 1
done
for c in A D; do for r in 1 4; do
  echo "config $c POLAR_HIP_RELAY=$r" >> $o
  POLAR_HIP_RELAY=$r timeout -k 10 300 python3 bench.py --config $c --steps 5 --warmup 2 --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" >> $o || exit 1
done; done
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 300 --timeout-method thread -k "bench or full or golden or mixed or ragged" 2>&1 | tail -5 >> $o
cat $o
