#!/bin/bash
# A/B of two builds of the library inside one gpurun call (same box)
export TMPDIR=/tmp
for rep in 1 2; do
for lib in libpolar_hip_old.so libpolar_hip.so; do
  for v in "10 4194304" "16 524288" "12 1048576"; do
    set -- $v
    POLAR_HIP_LIB=$PWD/polar_amd/$lib timeout -k 10 300 python3 tools/time_decode.py --M $1 --frames $2 --reps 3 2>&1 | grep "^M " | sed "s/^/$lib /" || exit 1
  done
done
done
