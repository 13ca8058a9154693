#!/bin/bash
# what the driver runs at round end: GPU tests, smoke, bench
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests.log 2>&1 || { tail -40 gpurun_out/gputests.log; exit 1; }
tail -3 gpurun_out/gputests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" || exit 1
timeout -k 10 600 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err || { tail -20 gpurun_out/bench_final.err; exit 1; }
cat gpurun_out/bench_final.json
