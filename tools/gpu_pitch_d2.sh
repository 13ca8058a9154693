#!/bin/bash
export TMPDIR=/tmp POLAR_HIP_PLACEMENT=0
mkdir -p gpurun_out
o=gpurun_out/r3_row_pitch_D_out.txt
: > $o
for side in out both; do
echo "config D side $side" >> $o
timeout -k 10 500 python3 tools/ab_inproc.py --M 20 --frames 65536 --rounds 2 --side $side --strides 0,4352,256,2304 2>&1 | grep -E "kernel_ms|rror" | sed 's/ pad 0 skew None hot skew None//' >> $o || { cat $o; exit 1; }
done
cat $o
