#!/bin/bash
# A/B of the experiment build (libpolar_hip_x.so) against the product library, same box, N = 2^16
# usage: tools/gpu_ab_lib.sh [frames] [extra time_decode args]
export TMPDIR=/tmp
mkdir -p gpurun_out
frames=${1:-1048576}
shift
for rep in 1 2; do
for lib in libpolar_hip.so libpolar_hip_x.so; do
  POLAR_HIP_LIB=$PWD/polar_amd/$lib timeout -k 10 300 python3 tools/time_decode.py --M 16 --frames $frames --reps 2 "$@" 2>&1 | grep "^M " | sed "s/^/$lib /" | tee -a gpurun_out/r3_ab.txt || exit 1
done
done
