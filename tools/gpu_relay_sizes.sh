#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 500 python3 -u tools/relay_sizes.py 2>&1 | grep -E "N=|rror" | tee gpurun_out/r3_relay_sizes.txt
