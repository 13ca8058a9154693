#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
o=gpurun_out/r3_relay_cohorts.txt
: > $o
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py tests/test_zz_no_alloc.py -x -q -m gpu --timeout 300 --timeout-method thread -k "relay or pitched or placement or mixed or no_alloc or static" 2>&1 | tail -4 >> $o || { cat $o; exit 1; }
for c in B A; do for r in "" 1 "" 1; do
  echo "config $c POLAR_HIP_RELAY=${r:-default}" >> $o
  if [ -n "$r" ]; then export POLAR_HIP_RELAY=$r; else unset POLAR_HIP_RELAY; fi
  POLAR_HIP_DEBUG=1 timeout -k 10 300 python3 bench.py --config $c --steps 5 --warmup 1 --cpu-seconds 0 2> gpurun_out/flip.err | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('launches_per_decode'), d['ber'])" >> $o || { cat $o; tail gpurun_out/flip.err; exit 1; }
  grep "relay stream" gpurun_out/flip.err | head -4 >> $o
done; done
unset POLAR_HIP_RELAY
timeout -k 10 300 python3 -u tools/soak_relay.py --seconds 60 --seed 5 2>&1 | tail -1 >> $o
cat $o
