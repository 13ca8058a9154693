#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1000 python tools/ber_sweep.py --M 14 16 --both --max-frames 1048576 --chunk 131072 > gpurun_out/ber_m14_m16.txt 2> gpurun_out/ber.err || { tail -20 gpurun_out/ber.err; exit 1; }
timeout -k 10 1000 python tools/ber_sweep.py --M 20 --both --max-frames 16384 --chunk 8192 > gpurun_out/ber_m20.txt 2> gpurun_out/ber20.err || { tail -20 gpurun_out/ber20.err; exit 1; }
tail -12 gpurun_out/ber_m20.txt
