#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests.log 2>&1 || { tail -40 gpurun_out/gputests.log; exit 1; }
tail -3 gpurun_out/gputests.log
for v in "16 2048" "16 8192" "16 32768" "16 65536" "16 131072" "16 524288" "10 4194304" "10 65536" "12 65536" "14 65536" "20 65536"; do
  set -- $v
  timeout -k 10 300 python3 tools/time_decode.py --M $1 --frames $2 --reps 3 2>&1 | grep "^M " || exit 1
done
