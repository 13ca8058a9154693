#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for mode in none same prio; do
  rm -rf gpurun_out/trace_E_set
  if [ $mode = none ]; then unset POLAR_HIP_SET_STREAMS; else export POLAR_HIP_SET_STREAMS=$mode; fi
  timeout -s KILL 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_E_set -- python3 bench.py --config E --steps 1 --warmup 1 --cpu-seconds 0 > gpurun_out/trace_E_set.json 2> gpurun_out/trace_E_set.err || { tail -5 gpurun_out/trace_E_set.err; exit 1; }
  python3 tools/trace_overlap.py gpurun_out/trace_E_set 28 | sed "s/^/polar_hip_decode_set, lane streams $mode: /"
  rm -rf gpurun_out/trace_E_set
done
