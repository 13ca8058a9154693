#!/bin/bash
# row pitch of the input for config D (N = 2^20, 2^16 frames) and config A (N = 2^10, 2^22 frames)
export TMPDIR=/tmp POLAR_HIP_PLACEMENT=0
mkdir -p gpurun_out
o=gpurun_out/r3_row_pitch_AD.txt
: > $o
echo "config D" >> $o
timeout -k 10 500 python3 tools/ab_inproc.py --M 20 --frames 65536 --rounds 2 --side in --strides 0,4352,256,1024,2304,4096,8448,12544,16640,1280,768 2>&1 | grep -E "kernel_ms|rror" | sed 's/ pad 0 skew None hot skew None//' >> $o || { cat $o; exit 1; }
echo "config A" >> $o
timeout -k 10 500 python3 tools/ab_inproc.py --M 10 --frames 4194304 --rounds 2 --side in --strides 0,4352,256,1024,2304,4096,8448,12544,16640,1280,768 2>&1 | grep -E "kernel_ms|rror" | sed 's/ pad 0 skew None hot skew None//' >> $o || { cat $o; exit 1; }
cat $o
