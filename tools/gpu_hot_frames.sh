#!/bin/bash
# hot levels at several batch sizes (sweeps of the 2048 resident wavefronts), (128, 6), one box
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r3_hot_frames.txt
: > $out
for frames in ${FRAMES:-262144 524288 1048576}; do
  for hot in ${HOTS:-0 1 2 3 4}; do
    POLAR_HIP_HOT_LEVELS=$hot timeout -k 10 300 python3 tools/time_decode.py --M 16 --frames $frames --fpw 128 --lds-level 6 --reps 2 2>&1 | grep "^M " | sed "s/^/hot $hot: /" | tee -a $out || exit 1
  done
done
