#!/bin/bash
# experiment: team form (4 wavefronts per workgroup, barrier before every pass over the input rows)
export TMPDIR=/tmp
mkdir -p gpurun_out
POLAR_HIP_TEAM=1 timeout -k 10 120 python3 tools/soak_gpu.py --seconds 40 > gpurun_out/team_soak.txt 2>&1 || { tail -5 gpurun_out/team_soak.txt; exit 1; }
tail -1 gpurun_out/team_soak.txt
for v in "16 32768 32 8" "16 65536 32 8" "16 524288 32 8" "16 131072 64 7" "16 524288 64 7" "20 65536 32 8" "20 65536 64 7" "10 4194304 32 8"; do
  set -- $v
  for team in 0 1; do
    if [ $team = 1 ]; then export POLAR_HIP_TEAM=1; else unset POLAR_HIP_TEAM; fi
    timeout -k 10 300 python3 tools/time_decode.py --M $1 --frames $2 --fpw $3 --lds-level $4 --reps 2 2>&1 | grep "^M " | sed "s/^/team=$team /" || exit 1
  done
done
