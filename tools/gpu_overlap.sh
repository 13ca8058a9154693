#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python3 tools/overlap_halves.py 2>&1 | grep -E "ms|rror|Trace" | tee gpurun_out/r3_overlap_halves.txt
