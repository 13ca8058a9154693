#!/bin/bash
# VALU instruction counts of the decode kernel on small all-LDS codes: instructions per node
export TMPDIR=/tmp
mkdir -p gpurun_out/r2b
for M in 4 6 8 10 12; do
  F=$((2048*128*4))
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES --output-format csv -d gpurun_out/r2b/m$M -- python3 tools/time_decode.py --M $M --frames $F --fpw 128 --lds-level 6 --reps 2 > gpurun_out/r2b/m$M.log 2>&1 || { echo fail $M; tail -5 gpurun_out/r2b/m$M.log; exit 1; }
  grep "^M " gpurun_out/r2b/m$M.log
  python3 - $M <<'PY'
import csv,glob,sys
from collections import defaultdict
M=sys.argv[1]
v=defaultdict(list)
for f in glob.glob("gpurun_out/r2b/m%s/**/*counter_collection.csv"%M, recursive=True):
    for r in csv.DictReader(open(f)):
        if "sc_decode" in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("M",M," ".join("%s=%.4g"%(k,max(x)) for k,x in sorted(v.items())))
PY
  rm -rf gpurun_out/r2b/m$M
done
