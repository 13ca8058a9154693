#!/bin/bash
PROG=$1; G=${2:-64}; L=${3:-7}; F=${4:-229376}; TAG=${5:-prog}
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="tools/time_program.py --prog $PROG --fpw $G --lds-level $L --frames $F --reps 2"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_IFETCH" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -- python3 $ARGS > $OUT/pmc$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/pmc$i.log; }
done
grep "^prog" $OUT/pmc1.log | tail -1
python3 tools/prof_summary.py $OUT
