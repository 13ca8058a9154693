#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for B in 0 3 5 7 9; do
  timeout -k 10 200 python3 tools/time_decode.py --M 16 --frames 524288 --fpw 128 --lds-level 6 --reps 2 --block-frozen $B 2>&1 | grep "^M " | sed "s/^/B=$B /" | tee -a gpurun_out/blockfrozen.txt
done
