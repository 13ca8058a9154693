#!/usr/bin/env python3
"""Small driver for rocprofv3: builds a resident batch like bench.py and runs the decode kernel
`--reps` times. Usage: rocprofv3 ... -- python3 tools/prof_run.py --frames F --fpw G --lds-level L"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=32768)
ap.add_argument("--fpw", type=int, default=0)
ap.add_argument("--lds-level", type=int, default=0)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
import torch  # noqa: E402
import polar_amd  # noqa: E402

dev = torch.device("cuda", 0)
sess = polar_amd.Session(0)
if a.fpw:
    sess.configure(a.fpw, a.lds_level, 0)
fro, prog = bench.make_code(polar_amd)
llr, u_true, _ = bench.generate(torch, polar_amd, sess, fro, prog, a.frames, 42, dev)
out = torch.empty((bench.K, a.frames), dtype=torch.int8, device=dev)
dec = polar_amd.PolarDecoder(sess)
for _ in range(a.reps):
    dec(out, llr, prog)
    print("kernel_ms", sess.last_decode_ms(), flush=True)
