#!/bin/bash
# round 2, first GPU session: VALU issue-rate microbench, GPU tests, per-node profiles of narrow variants, PMC of the shipped build
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 120 tools/ubench/valu_rate > gpurun_out/valu_rate.txt 2>&1 || { echo valu_rate failed; tail -5 gpurun_out/valu_rate.txt; exit 1; }
cat gpurun_out/valu_rate.txt
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests.log 2>&1 || { tail -40 gpurun_out/gputests.log; exit 1; }
tail -3 gpurun_out/gputests.log
for cfg in "32 8 65536" "32 7 65536" "128 6 262144"; do
  set -- $cfg
  POLAR_HIP_LIB=$PWD/polar_amd/libpolar_hip_prof.so timeout -k 10 300 python3 tools/time_decode.py --M 16 --frames $3 --fpw $1 --lds-level $2 --reps 1 > gpurun_out/nodeprof_r2_$1_$2.txt 2>&1 || { echo nodeprof $cfg failed; tail -5 gpurun_out/nodeprof_r2_$1_$2.txt; exit 1; }
  python3 tools/nodeprof_summary.py gpurun_out/nodeprof_r2_$1_$2.txt $2 12
done
bash tools/prof_pmc.sh 16 2 128 6 r2a_128_6
