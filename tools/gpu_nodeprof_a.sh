#!/bin/bash
# per-node cycle stamps for config A (N = 2^10)
export TMPDIR=/tmp
mkdir -p gpurun_out
POLAR_HIP_LIB=$PWD/polar_amd/libpolar_hip_prof.so timeout -k 10 300 python3 tools/time_decode.py --M 10 --frames 4194304 --reps 1 > gpurun_out/r3_nodeprof_A.txt 2>&1 || exit 1
tail -1 gpurun_out/r3_nodeprof_A.txt
