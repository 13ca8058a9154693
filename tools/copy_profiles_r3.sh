#!/bin/bash
# gpurun_out/prof_r3_* and bench_r3_*.json (tools/gpu_bench_profiles_r3.sh) -> profiles/
for c in B A D; do
  s=gpurun_out/prof_r3_$c
  cp $s/kernel_stats.csv profiles/r3_bench_${c}_kernel_stats.csv
  cp $s/summary.txt profiles/r3_bench_${c}_summary.txt
  cp $s/bench_under_trace.json profiles/r3_bench_${c}_under_trace.json
done
cp gpurun_out/prof_r3_B/hbm_traffic.json profiles/hbm_traffic.json
cp gpurun_out/prof_r3_A/hbm_traffic.json profiles/hbm_traffic_A.json
cp gpurun_out/prof_r3_D/hbm_traffic.json profiles/hbm_traffic_D.json
cp gpurun_out/prof_r3_E/kernel_stats.csv profiles/r3_bench_E_kernel_stats.csv
for c in B A D E; do cp gpurun_out/bench_r3_$c.json profiles/r3_bench_$c.json; done
