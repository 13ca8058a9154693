#!/bin/bash
# rocprofv3 evidence for the bench line: kernel trace + stats, then PMC passes, each in a run of
# its own (MI355X_MICROARCH.md: FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2, at most 8 SQ counters).
# usage: tools/prof_bench.sh TAG [bench args...]      -> gpurun_out/prof_TAG/
TAG=${1:-r3}; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 3 --warmup 1 --cpu-seconds 0 $*"
timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/bench_under_trace.json 2> $OUT/trace.err || { echo "trace pass failed"; tail -5 $OUT/trace.err; exit 1; }
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout -s KILL 500 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -- python3 $ARGS > $OUT/bench_under_pmc$i.json 2> $OUT/pmc$i.err || { echo "pmc pass $i failed"; tail -5 $OUT/pmc$i.err; exit 1; }
done
python3 tools/prof_bench_summary.py $OUT $* > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
rm -rf $OUT/trace $OUT/pmc[0-9]*/
