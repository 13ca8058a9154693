#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for k in 2 4; do
  rm -rf gpurun_out/trace_E_$k
  timeout -s KILL 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_E_$k -- python3 bench.py --config E --steps 1 --warmup 1 --cpu-seconds 0 --streams $k > gpurun_out/trace_E_$k.json 2> gpurun_out/trace_E_$k.err || { tail -5 gpurun_out/trace_E_$k.err; exit 1; }
  python3 tools/trace_overlap.py gpurun_out/trace_E_$k 28 | sed "s/^/streams $k: /" | tee -a gpurun_out/r3_bench_E_streams.txt
  rm -rf gpurun_out/trace_E_$k
done
