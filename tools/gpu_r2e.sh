#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests.log 2>&1 || { tail -40 gpurun_out/gputests.log; exit 1; }
tail -3 gpurun_out/gputests.log
timeout -k 10 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_r2e.json 2> gpurun_out/bench_r2e.err || { tail -20 gpurun_out/bench_r2e.err; exit 1; }
cat gpurun_out/bench_r2e.json
timeout -k 10 600 python bench.py --steps 3 --warmup 1 --include-extract --cpu-seconds 0 > gpurun_out/bench_r2e_x.json 2> gpurun_out/bench_r2e_x.err || { tail -20 gpurun_out/bench_r2e_x.err; exit 1; }
cat gpurun_out/bench_r2e_x.json
