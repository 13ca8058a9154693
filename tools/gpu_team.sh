#!/bin/bash
# experiment: team forms (T wavefronts per workgroup, barrier before every pass over the input rows)
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { # M frames fpw rl team
  if [ $5 -gt 1 ]; then export POLAR_HIP_TEAM=$5; else unset POLAR_HIP_TEAM; fi
  timeout -k 10 300 python3 tools/time_decode.py --M $1 --frames $2 --fpw $3 --lds-level $4 --reps 2 2>&1 | grep "^M " | sed "s/^/team=$5 /" || exit 1
}
for t in 4 8; do POLAR_HIP_TEAM=$t timeout -k 10 120 python3 tools/soak_gpu.py --seconds 25 2>&1 | tail -1 || exit 1; done
run 16 524288 32 8 8
run 16 524288 32 8 4
run 16 524288 64 7 2
run 16 524288 64 7 4
run 16 524288 128 6 4
run 16 524288 128 6 1
run 16 1048576 64 7 4
run 16 1048576 128 6 1
run 16 524288 16 9 8
run 16 524288 16 9 1
run 16 32768 16 9 8
run 16 32768 16 9 1
run 16 32768 32 8 8
run 16 32768 32 8 4
run 20 65536 32 8 8
run 10 4194304 64 7 4
run 10 4194304 128 6 4
run 10 4194304 128 6 1
