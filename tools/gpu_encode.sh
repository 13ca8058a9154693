#!/bin/bash
# encoder / systematic-extract tests, then timings of the tile passes against the register-only passes
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread -k "encoders or systematic or sweep_harness or simulate or extract" > gpurun_out/enc_tests.log 2>&1 || { tail -40 gpurun_out/enc_tests.log; exit 1; }
tail -3 gpurun_out/enc_tests.log
for M in 16 10 12 20; do
  F=$(( (1 << 35) >> M ))
  timeout -k 10 300 python3 tools/time_encode.py --M $M --frames $F || exit 1
  POLAR_HIP_PACK_C=5 timeout -k 10 300 python3 tools/time_encode.py --M $M --frames $F || exit 1
done
timeout -k 10 300 python3 tools/time_extract.py || exit 1
POLAR_HIP_PACK_C=5 timeout -k 10 300 python3 tools/time_extract.py || exit 1
