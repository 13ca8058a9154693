#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
POLAR_HIP_LIB=$PWD/polar_amd/libpolar_hip_prof.so timeout -k 10 600 python3 tools/time_decode.py --M 20 --frames 65536 --fpw 32 --lds-level 8 --reps 1 > gpurun_out/nodeprof_r2_D.txt 2>&1 || { tail -5 gpurun_out/nodeprof_r2_D.txt; exit 1; }
python3 tools/nodeprof_summary.py gpurun_out/nodeprof_r2_D.txt 8 30
