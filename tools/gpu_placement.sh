#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 800 python3 tools/placement_probe4.py "$@" 2>&1 | grep -E "ms$|rror" | tee -a gpurun_out/r3_placement_pieces.txt
