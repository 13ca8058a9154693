#!/bin/bash
# experiment: teams of four for 16- and 8-frame groups in the small-batch regime
export TMPDIR=/tmp
run() { # M frames fpw rl team
  timeout -k 10 300 python3 tools/time_decode.py --M $1 --frames $2 --fpw $3 --lds-level $4 --team $5 --reps 2 2>&1 | grep "^M " | sed "s/^/team=$5 /" || exit 1
}
for f in 4096 8192 16384 24576; do
  run 16 $f 16 9 1
  run 16 $f 16 9 4
  run 16 $f 8 10 1
  run 16 $f 8 10 4
  run 16 $f 32 8 4
done
run 20 16384 16 9 1
run 20 16384 16 9 4
run 12 16384 16 9 1
run 12 16384 16 9 4
