#!/bin/bash
# GPU regression + timing of a few (frames per wave, LDS level) variants.
# usage: tools/run_gpu_check.sh "128 5" "64 7" ...
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests.log 2>&1 || { tail -30 gpurun_out/gputests.log; exit 1; }
tail -3 gpurun_out/gputests.log
for cfg in "$@"; do
  set -- $cfg
  timeout -k 10 200 python3 tools/time_decode.py --M 16 --sweeps 2 --fpw $1 --lds-level $2 --reps 2 | grep "^M " | tee -a gpurun_out/timing.txt
done
