#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python3 tools/placement_probe3.py 2>&1 | grep -E "ms$|scratch 0x|rror" | tee -a gpurun_out/r3_placement.txt
