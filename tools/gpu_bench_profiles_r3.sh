#!/bin/bash
# bench lines + rocprofv3 evidence for configs B (headline), A, D and E on the final library
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r3_E
bash tools/prof_bench.sh r3_B || exit 1
bash tools/prof_bench.sh r3_A --config A || exit 1
bash tools/prof_bench.sh r3_D --config D || exit 1
timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3_E/trace -- python3 bench.py --config E --steps 3 --warmup 1 --cpu-seconds 0 > gpurun_out/prof_r3_E/bench_under_trace.json 2> gpurun_out/prof_r3_E/trace.err || { tail -5 gpurun_out/prof_r3_E/trace.err; exit 1; }
cp $(ls gpurun_out/prof_r3_E/trace/*/*kernel_stats.csv | head -1) gpurun_out/prof_r3_E/kernel_stats.csv && rm -rf gpurun_out/prof_r3_E/trace
for c in B A D E; do
  timeout -k 10 600 python bench.py --config $c --steps 5 --warmup 1 > gpurun_out/bench_r3_$c.json 2> gpurun_out/bench_r3_$c.err || { tail -20 gpurun_out/bench_r3_$c.err; exit 1; }
  cut -c1-400 gpurun_out/bench_r3_$c.json
done
