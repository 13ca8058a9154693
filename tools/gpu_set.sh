#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity.py -x -q --timeout 250 --timeout-method thread -k "mixed_sizes" > gpurun_out/r3_set_test.log 2>&1 || { tail -30 gpurun_out/r3_set_test.log; exit 1; }
tail -2 gpurun_out/r3_set_test.log
for k in 0; do
  timeout -k 10 600 python bench.py --config E --steps 5 --warmup 1 --cpu-seconds 0 --streams $k > gpurun_out/bench_E_s$k.json 2> gpurun_out/bench_E_s$k.err || { tail -20 gpurun_out/bench_E_s$k.err; exit 1; }
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/bench_E_s$k.json'))
print('streams $k: %.1f Gbit/s, %.2f ms per step, set GPU time %.2f ms, frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))" | tee -a gpurun_out/r3_bench_E_streams.txt
done
rm -rf gpurun_out/trace_E_set
timeout -s KILL 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_E_set -- python3 bench.py --config E --steps 1 --warmup 1 --cpu-seconds 0 > gpurun_out/trace_E_set.json 2> gpurun_out/trace_E_set.err || { tail -5 gpurun_out/trace_E_set.err; exit 1; }
python3 tools/trace_overlap.py gpurun_out/trace_E_set 28 | sed "s/^/polar_hip_decode_set: /" | tee -a gpurun_out/r3_bench_E_streams.txt
rm -rf gpurun_out/trace_E_set
