#!/bin/bash
# bench.py kernel time under A/B settings, several processes each (process-to-process variance is large)
for i in 1 2 3; do
  for cfg in "hip -" "hip 0" "hip_x0 -" "hip_x0 0"; do
    set -- $cfg
    if [ "$2" = "-" ]; then unset POLAR_HIP_KEEP_BELOW; else export POLAR_HIP_KEEP_BELOW=$2; fi
    ms=$(POLAR_HIP_LIB=$PWD/polar_amd/libpolar_$1.so timeout -k 10 300 python3 bench.py --cpu-seconds 0 --steps 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f ms kernel, %.1f Gbit/s' % (d['roofline']['kernel_ms'], d['value']))")
    echo "$1 keep_below=$2: $ms"
  done
done
