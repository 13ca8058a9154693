#!/bin/bash
# kernel time per call over a long run of one process (clock / temperature drift), N = 2^16, 2^20 frames
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 400 python3 tools/time_decode.py --M 16 --frames 1048576 --reps 24 2>&1 | grep -E "kernel|^M " | tee gpurun_out/r3_drift.txt
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temp" | head -12 | tee -a gpurun_out/r3_drift.txt
