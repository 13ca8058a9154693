#!/bin/bash
export TMPDIR=/tmp
run() { # M frames fpw rl team
  if [ $5 -gt 1 ]; then export POLAR_HIP_TEAM=$5; else unset POLAR_HIP_TEAM; fi
  timeout -k 10 300 python3 tools/time_decode.py --M $1 --frames $2 --fpw $3 --lds-level $4 --reps 2 2>&1 | grep "^M " | sed "s/^/team=$5 /" || exit 1
}
run 16 262144 128 6 1
run 16 262144 64 7 4
run 16 393216 128 6 1
run 16 393216 64 7 4
for M in 14 12 10; do
  run $M 131072 64 7 1
  run $M 131072 64 7 4
  run $M 65536 32 8 1
  run $M 65536 32 8 4
  run $M 32768 32 8 1
  run $M 32768 32 8 4
done
run 20 32768 32 8 4
run 20 32768 32 8 1
run 20 32768 16 9 1
run 20 32768 16 9 8
