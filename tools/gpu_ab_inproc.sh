#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 800 python3 tools/ab_inproc.py "$@" 2>&1 | grep -E "^round|Error|error" | tee gpurun_out/r3_ab_inproc.txt
