#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 400 python3 -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -k "float" > gpurun_out/r3_float_tests.log 2>&1 || { tail -30 gpurun_out/r3_float_tests.log; exit 1; }
tail -2 gpurun_out/r3_float_tests.log
: > gpurun_out/r3_float_rate.txt
timeout -k 10 200 python3 tools/time_float.py --M 12 --frames 262144 2>&1 | grep "^M " | tee -a gpurun_out/r3_float_rate.txt || exit 1
timeout -k 10 200 python3 tools/time_float.py --M 16 --frames 16384 2>&1 | grep "^M " | tee -a gpurun_out/r3_float_rate.txt || exit 1
timeout -k 10 300 python3 tools/time_float.py --M 16 --frames 131072 2>&1 | grep "^M " | tee -a gpurun_out/r3_float_rate.txt || exit 1
