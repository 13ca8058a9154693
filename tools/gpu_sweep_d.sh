#!/bin/bash
# config D (N = 2^20) and config A (N = 2^10): kernel time per variant
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/sweep_d.txt
: > $OUT
for v in "16 9 32768" "32 8 32768" "64 7 32768" "128 6 32768" "128 6 65536" "64 7 65536" "32 8 65536" "256 5 65536"; do
  set -- $v
  timeout -k 10 300 python3 tools/time_decode.py --M 20 --frames $3 --fpw $1 --lds-level $2 --reps 2 2>&1 | grep "^M " >> $OUT || { echo "failed $v"; tail -3 $OUT; exit 1; }
done
for v in "128 6" "256 5" "256 4" "64 7" "64 6" "128 5"; do
  set -- $v
  timeout -k 10 300 python3 tools/time_decode.py --M 10 --frames 4194304 --fpw $1 --lds-level $2 --reps 3 2>&1 | grep "^M " >> $OUT || { echo "failed $v"; exit 1; }
done
cat $OUT
