#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 500 python3 -u tools/soak_relay.py --seconds 240 2>&1 | tee gpurun_out/r3_soak_relay.txt | tail -8
