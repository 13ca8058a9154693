#!/bin/bash
# hot levels with every subtree below level B replaced by a leaf (no LDS-level work): (128, 6)
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r3_hot_blockfrozen.txt
: > $out
for bf in 7 9; do
for frames in 262144 1048576; do
  for hot in 0 3; do
    POLAR_HIP_HOT_LEVELS=$hot timeout -k 10 300 python3 tools/time_decode.py --M 16 --frames $frames --fpw 128 --lds-level 6 --block-frozen $bf --reps 2 2>&1 | grep "^M " | sed "s/^/bf $bf hot $hot: /" | tee -a $out || exit 1
  done
done
done
