#!/bin/bash
export TMPDIR=/tmp
for c in 4 5 6; do echo "PACK_C=$c"; POLAR_HIP_PACK_C=$c timeout -k 10 200 python3 tools/time_extract.py || exit 1; done
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "packed or systematic or simulate" --timeout 300 --timeout-method thread 2>&1 | tail -3
