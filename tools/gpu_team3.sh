#!/bin/bash
# after making the team forms the default for 32- and 64-frame groups: GPU suite, soak, default-pick sweep
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests.log 2>&1 || { tail -40 gpurun_out/gputests.log; exit 1; }
tail -3 gpurun_out/gputests.log
timeout -k 10 200 python3 tools/soak_gpu.py --seconds 90 > gpurun_out/soak.txt 2>&1 || { tail -5 gpurun_out/soak.txt; exit 1; }
tail -1 gpurun_out/soak.txt
for f in 2048 8192 16384 32768 65536 131072 262144 524288 1048576; do
  timeout -k 10 300 python3 tools/time_decode.py --M 16 --frames $f --reps 2 2>&1 | grep "^M " || exit 1
done
timeout -k 10 300 python3 tools/time_decode.py --M 20 --frames 65536 --reps 2 2>&1 | grep "^M " || exit 1
timeout -k 10 300 python3 tools/time_decode.py --M 10 --frames 4194304 --reps 2 2>&1 | grep "^M " || exit 1
