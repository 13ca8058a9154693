#!/bin/bash
# per-node cycle profile of the narrow-group variants in the small-batch (latency) regime
export TMPDIR=/tmp
mkdir -p gpurun_out
for cfg in "4 11 2048" "8 10 8192" "32 8 32768"; do
  set -- $cfg
  POLAR_HIP_LIB=$PWD/polar_amd/libpolar_hip_prof.so timeout -k 10 300 python3 tools/time_decode.py --M 16 --frames $3 --fpw $1 --lds-level $2 --reps 1 > gpurun_out/nodeprof_narrow_$1_$2.txt 2>&1 || { echo nodeprof $cfg failed; tail -5 gpurun_out/nodeprof_narrow_$1_$2.txt; exit 1; }
  python3 tools/nodeprof_summary.py gpurun_out/nodeprof_narrow_$1_$2.txt $2 40
done
