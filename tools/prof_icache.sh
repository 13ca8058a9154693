#!/bin/bash
M=${1:-6}; F=${2:-33554432}; G=${3:-64}; L=${4:-6}; TAG=${5:-ic}
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="tools/time_decode.py --M $M --frames $F --fpw $G --lds-level $L --reps 2"
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_IFETCH SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -- python3 $ARGS > $OUT/pmc$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/pmc$i.log; }
done
python3 tools/prof_summary.py $OUT
