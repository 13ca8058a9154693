#!/bin/bash
# A/B inside one gpurun call: shipped library vs experiment build, given kernel shapes
export TMPDIR=/tmp
for rep in 1 2; do
for lib in libpolar_hip.so libpolar_hip_x.so; do
  for v in "16 524288 64 7 4" "16 1048576 64 7 4" "16 1048576 128 6 1" "16 131072 64 7 4"; do
    set -- $v
    POLAR_HIP_LIB=$PWD/polar_amd/$lib timeout -k 10 300 python3 tools/time_decode.py --M $1 --frames $2 --fpw $3 --lds-level $4 --team $5 --reps 2 2>&1 | grep "^M " | sed "s/^/$lib team=$5 /" || exit 1
  done
done
done
