#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for cfg in "3 680" "4 512" "4 768" "4 1024" "4 2048" "2 2048"; do
  set -- $cfg
  timeout -k 10 600 python bench.py --config E --steps 5 --warmup 1 --cpu-seconds 0 --streams $1 --lane-waves $2 > gpurun_out/bench_E_x.json 2> gpurun_out/bench_E_x.err || { tail -20 gpurun_out/bench_E_x.err; exit 1; }
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/bench_E_x.json'))
print('streams $1 waves $2: %.1f Gbit/s, %.2f ms per step; %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel'][-150:]))" | tee -a gpurun_out/r3_bench_E_streams.txt
done
