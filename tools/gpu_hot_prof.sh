#!/bin/bash
# per-node cycle stamps (profile build) with 0 and 3 hot scratch levels, same box
export TMPDIR=/tmp
mkdir -p gpurun_out
for hot in 0 4; do
  POLAR_HIP_HOT_LEVELS=$hot POLAR_HIP_LIB=$PWD/polar_amd/libpolar_hip_prof.so timeout -k 10 300 python3 tools/time_decode.py --M 16 --frames 524288 --fpw 128 --lds-level 6 --reps 1 > gpurun_out/r3_nodeprof_hot$hot.txt 2>&1 || exit 1
  tail -1 gpurun_out/r3_nodeprof_hot$hot.txt
done
