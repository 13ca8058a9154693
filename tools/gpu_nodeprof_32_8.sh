#!/bin/bash
# per-node cycle stamps of the (32, 8) team kernel: latency regime (32 Ki frames) and full load (2^19)
export TMPDIR=/tmp
mkdir -p gpurun_out
for f in 32768 524288; do
  POLAR_HIP_LIB=$PWD/polar_amd/libpolar_hip_prof.so timeout -k 10 300 python3 tools/time_decode.py --M 16 --frames $f --fpw 32 --lds-level 8 --team 4 --reps 1 > gpurun_out/r3_nodeprof_32_8_$f.txt 2>&1 || exit 1
  tail -1 gpurun_out/r3_nodeprof_32_8_$f.txt
done
