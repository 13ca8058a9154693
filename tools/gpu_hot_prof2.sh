#!/bin/bash
# per-node cycle stamps (profile build), few resident wavefronts (no queueing): do the hot levels hit a cache?
export TMPDIR=/tmp
mkdir -p gpurun_out
for hot in 0 3; do
  POLAR_HIP_DEBUG=1 POLAR_HIP_HOT_LEVELS=$hot POLAR_HIP_LIB=$PWD/polar_amd/libpolar_hip_prof.so timeout -k 10 300 python3 tools/time_decode.py --M 16 --frames 32768 --fpw 128 --lds-level 6 --waves 256 --reps 1 > gpurun_out/r3_nodeprof_w256_hot$hot.txt 2>&1 || exit 1
  tail -1 gpurun_out/r3_nodeprof_w256_hot$hot.txt
done
