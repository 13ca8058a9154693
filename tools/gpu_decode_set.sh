#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity.py -x -q --timeout 250 --timeout-method thread -k "mixed_sizes" > gpurun_out/r3_set_test.log 2>&1 || { tail -30 gpurun_out/r3_set_test.log; exit 1; }
tail -1 gpurun_out/r3_set_test.log
timeout -k 10 600 python bench.py --config E --steps 5 --warmup 1 --cpu-seconds 0 > gpurun_out/bench_E_s0.json 2> gpurun_out/bench_E_s0.err || { tail -20 gpurun_out/bench_E_s0.err; exit 1; }
cut -c1-250 gpurun_out/bench_E_s0.json
