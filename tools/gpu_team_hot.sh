#!/bin/bash
# hot levels in the team kernels (experiment build -DPOLAR_TEAM_HOT=1) against the product library
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r3_team_hot.txt
: > $out
run() { # label, args...
  for rep in 1 2; do
  for lib in libpolar_hip.so libpolar_hip_x.so; do
    POLAR_HIP_LIB=$PWD/polar_amd/$lib timeout -k 10 400 python3 tools/time_decode.py "$@" --reps 2 2>&1 | grep "^M " | sed "s/^/$lib /" | tee -a $out || exit 1
  done
  done
}
run --M 16 --frames 1048576 --fpw 64 --lds-level 7 --team 4
run --M 16 --frames 1048576 --fpw 32 --lds-level 8 --team 4
run --M 20 --frames 65536
