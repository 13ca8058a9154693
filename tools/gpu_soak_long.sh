#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 500 python3 -u tools/soak_gpu.py --seconds 420 --seed 77 --min-m 3 --max-m 12 > gpurun_out/r3_soak_small_long.txt 2>&1 || { tail -5 gpurun_out/r3_soak_small_long.txt; exit 1; }
tail -1 gpurun_out/r3_soak_small_long.txt
timeout -k 10 600 python3 -u tools/soak_gpu.py --seconds 480 --seed 78 --min-m 12 --max-m 15 > gpurun_out/r3_soak_large_long.txt 2>&1 || { tail -5 gpurun_out/r3_soak_large_long.txt; exit 1; }
tail -1 gpurun_out/r3_soak_large_long.txt
