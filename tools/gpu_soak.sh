#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 200 python3 -u tools/soak_gpu.py --seconds 100 --seed 31 --min-m 3 --max-m 12 > gpurun_out/r3_soak_small.txt 2>&1 || { tail -5 gpurun_out/r3_soak_small.txt; exit 1; }
tail -1 gpurun_out/r3_soak_small.txt
timeout -k 10 400 python3 -u tools/soak_gpu.py --seconds 240 --seed 32 --min-m 12 --max-m 15 > gpurun_out/r3_soak_large.txt 2>&1 || { tail -5 gpurun_out/r3_soak_large.txt; exit 1; }
tail -1 gpurun_out/r3_soak_large.txt
