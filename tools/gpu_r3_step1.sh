#!/bin/bash
# float path tests on the product library, parity of the experiment library, A/B timing
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity.py -x -q --timeout 200 --timeout-method thread -k "float or dropin or native" > gpurun_out/r3_float_tests.log 2>&1 || { tail -30 gpurun_out/r3_float_tests.log; exit 1; }
tail -2 gpurun_out/r3_float_tests.log
POLAR_HIP_LIB=$PWD/polar_amd/libpolar_hip_x.so timeout -k 10 500 python3 -u -m pytest tests/test_gpu_parity.py -x -q --timeout 200 --timeout-method thread -k "wide_variants or bench_grid or golden_digests or all_variants or persistent" > gpurun_out/r3_x_parity.log 2>&1 || { tail -30 gpurun_out/r3_x_parity.log; exit 1; }
tail -2 gpurun_out/r3_x_parity.log
: > gpurun_out/r3_ab.txt
bash tools/gpu_ab_lib.sh 1048576
