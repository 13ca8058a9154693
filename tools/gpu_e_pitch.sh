#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
o=gpurun_out/r3_row_pitch_E.txt
: > $o
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 300 --timeout-method thread -k "mixed_sizes or pitched" 2>&1 | tail -5 >> $o || { cat $o; exit 1; }
for f in "" "--contiguous-llr" "" "--contiguous-llr"; do
  echo "config E $f" >> $o
  timeout -k 10 400 python3 bench.py --config E $f --steps 5 --warmup 1 --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" >> $o || { cat $o; exit 1; }
done
cat $o
