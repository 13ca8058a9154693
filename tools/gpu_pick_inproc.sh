#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python3 tools/pick_sweep_inproc.py "$@" 2>&1 | grep -E "^M |rror|Trace" | tee -a gpurun_out/r3_pick_sweep.txt
