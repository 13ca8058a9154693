#!/bin/bash
# config E with the launch set spread over 1, 2, 4, 8 handles / streams
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/r3_bench_E_streams.txt
for k in 1 2 4 8; do
  timeout -k 10 600 python bench.py --config E --steps 5 --warmup 1 --cpu-seconds 0 --streams $k > gpurun_out/bench_E_s$k.json 2> gpurun_out/bench_E_s$k.err || { tail -20 gpurun_out/bench_E_s$k.err; exit 1; }
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/bench_E_s$k.json'))
print('streams $k: %.1f Gbit/s, %.2f ms per step, set GPU time %.2f ms, frac %.4f; %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['kernel'][-150:]))" | tee -a gpurun_out/r3_bench_E_streams.txt
done
