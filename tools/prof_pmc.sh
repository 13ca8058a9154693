#!/bin/bash
# rocprofv3 passes for the decode kernel; one counter group per run (MI355X_MICROARCH.md: PMC slots)
# usage: tools/prof_pmc.sh M SWEEPS FPW LDS_LEVEL TAG
M=${1:-16}; S=${2:-2}; G=${3:-64}; L=${4:-7}; TAG=${5:-r1}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="tools/time_decode.py --M $M --sweeps $S --fpw $G --lds-level $L --reps 2"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_IFETCH SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -- python3 $ARGS > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -5 $OUT/pmc$i.log; }
done
grep "^M " $OUT/trace.log > $OUT/summary.txt
python3 tools/prof_summary.py $OUT >> $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
# keep the summary and the kernel stats table, drop the bulky raw traces (gpurun_out/ is capped at 64 MiB)
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv 2>/dev/null || true
rm -rf $OUT/trace $OUT/pmc[0-9]*/
