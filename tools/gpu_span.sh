#!/bin/bash
# when do the 2048 wavefronts of the headline launch start and end? (-DPOLAR_SPAN build)
export TMPDIR=/tmp
mkdir -p gpurun_out
POLAR_HIP_PLACEMENT=0 POLAR_HIP_LIB=$PWD/polar_amd/libpolar_hip_x.so timeout -k 10 300 python3 tools/time_decode.py --M 16 --frames 1048576 --reps 3 2>&1 | grep -E "wavefront life|index / 64|^M " | tee gpurun_out/r3_wave_spans.txt
