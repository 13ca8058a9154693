#!/bin/bash
# kernel time of every built variant over batch sizes: the data behind pick_variant's table
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/pick_sweep.txt
: > $OUT
for M in 16; do
for F in 2048 8192 32768 131072 524288; do
  for v in "4 11" "8 10" "16 9" "32 8" "64 7" "128 6" "32 7" "64 6" "16 8" "256 5" "128 5"; do
    set -- $v
    timeout -k 10 120 python3 tools/time_decode.py --M $M --frames $F --fpw $1 --lds-level $2 --reps 2 2>&1 | grep "^M " >> $OUT || { echo "failed M $M F $F $v"; exit 1; }
  done
done
done
for M in 10 12 14; do
for F in 4096 65536 1048576; do
  for v in "4 11" "16 9" "32 8" "64 7" "128 6" "256 5"; do
    set -- $v
    timeout -k 10 120 python3 tools/time_decode.py --M $M --frames $F --fpw $1 --lds-level $2 --reps 2 2>&1 | grep "^M " >> $OUT || { echo "failed M $M F $F $v"; exit 1; }
  done
done
done
cat $OUT
