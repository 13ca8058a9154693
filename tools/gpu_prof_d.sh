#!/bin/bash
# config D again (the default pick is now the team form of (32, 8)): profile + bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/prof_bench.sh r2_D --config D || exit 1
timeout -k 10 600 python bench.py --config D --steps 5 --warmup 1 > gpurun_out/bench_r2_D.json 2> gpurun_out/bench_r2_D.err || { tail -20 gpurun_out/bench_r2_D.err; exit 1; }
cat gpurun_out/bench_r2_D.json
