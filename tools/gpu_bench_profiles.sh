#!/bin/bash
# bench lines + rocprofv3 evidence for configs B (headline), A and D
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/prof_bench.sh r2_B || exit 1
bash tools/prof_bench.sh r2_A --config A || exit 1
bash tools/prof_bench.sh r2_D --config D || exit 1
for c in A D; do
  timeout -k 10 600 python bench.py --config $c --steps 5 --warmup 1 > gpurun_out/bench_r2_$c.json 2> gpurun_out/bench_r2_$c.err || { tail -20 gpurun_out/bench_r2_$c.err; exit 1; }
  cat gpurun_out/bench_r2_$c.json
done
