#!/bin/bash
# which row pitch of the caller's tensors suits the decode kernel: input only, output only, both
export TMPDIR=/tmp POLAR_HIP_PLACEMENT=0
mkdir -p gpurun_out
o=gpurun_out/r3_row_pitch.txt
: > $o
for side in in out both; do
  echo "side $side" >> $o
  timeout -k 10 400 python3 tools/ab_inproc.py --rounds 2 --side $side --strides 0,4352,256,1024,2304,4096,8448,12544,16640,33024,1280 2>&1 | grep -E "kernel_ms|rror" | sed 's/ pad 0 skew None hot skew None//' >> $o || exit 1
done
cat $o
