#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests.log 2>&1 || { tail -40 gpurun_out/gputests.log; exit 1; }
tail -3 gpurun_out/gputests.log
timeout -k 10 900 python tools/ber_sweep.py --M 14 16 --both --max-frames 65536 --chunk 16384 > gpurun_out/ber_m14_m16.txt 2> gpurun_out/ber.err || { tail -20 gpurun_out/ber.err; exit 1; }
tail -50 gpurun_out/ber_m14_m16.txt
