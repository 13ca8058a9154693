#!/bin/bash
export TMPDIR=/tmp
for s in 1 2 3 4 6 8; do
  timeout -k 10 300 python3 tools/time_decode.py --M 16 --frames $((262144*s)) --fpw 128 --lds-level 6 --reps 3 2>&1 | grep "^M " || exit 1
done
