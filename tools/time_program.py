#!/usr/bin/env python3
"""Time an arbitrary byte-code program (streaming micro-benchmarks of single node types).
  python3 tools/time_program.py --prog 16,0,3,9,255 --frames 229376 --fpw 64 --lds-level 7"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--prog", required=True)
ap.add_argument("--frames", type=int, default=229376)
ap.add_argument("--fpw", type=int, default=64)
ap.add_argument("--lds-level", type=int, default=7)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
import torch, polar_amd
prog = np.array([int(x) for x in a.prog.split(",")], np.uint8)
rows = polar_amd.lib.polar_program_rows(prog.ctypes.data, prog.size)
assert rows >= 0, polar_amd.lib.polar_hip_last_error()
N = 1 << int(prog[0])
sess = polar_amd.Session(0); sess.configure(a.fpw, a.lds_level, 0)
llr = torch.randint(-100, 100, (N, a.frames), dtype=torch.int8, device="cuda")
out = torch.empty((max(rows, 1), a.frames), dtype=torch.int8, device="cuda")
dec = polar_amd.PolarDecoder(sess)
ms = []
for _ in range(a.reps):
    dec(out, llr, prog); ms.append(sess.last_decode_ms())
print("prog %s rows %d frames %d: kernel_ms %.3f  (N*frames = %.1f GB -> %.2f TB/s per N)" % (a.prog, rows, a.frames, min(ms), N * a.frames / 1e9, N * a.frames / min(ms) / 1e9), flush=True)
