#!/usr/bin/env python3
"""Kernel time of the candidate shapes over batch sizes, all in ONE process (placements held fixed per shape):
the data behind pick_variant's thresholds after round 3's kernel changes.
  python3 tools/pick_sweep_inproc.py --M 16"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=16)
ap.add_argument("--frames", default="16384,24576,32768,49152,65536,98304,131072,196608,262144,393216,524288,786432")
a = ap.parse_args()
import torch
import polar_amd

os.environ["POLAR_HIP_PLACEMENT"] = "0"
N = 1 << a.M
frames = [int(x) for x in a.frames.split(",")]
fro = np.zeros(N, np.uint8)
polar_amd.PolarCodeConst0()(fro, a.M, N // 2, polar_amd.testbench_probability())
prog = np.zeros(2 * N + 1, np.uint8)
n = polar_amd.PolarCompiler()(prog, fro, a.M)
prog = prog[:n].copy()
sigma = float(np.sqrt(1.0 / (2.0 * 10.0 ** (-1.0 / 10.0))))
g = torch.Generator(device="cuda").manual_seed(1)
maxf = max(frames)
flat = torch.empty(N * maxf, dtype=torch.int8, device="cuda")
step = 1 << 28
for r in range(0, N * maxf, step):
    m = min(step, N * maxf - r)
    y = 1.0 + sigma * torch.randn(m, generator=g, device="cuda")
    flat[r:r + m] = torch.clamp(torch.round(y * (2.0 / sigma**2)), -128, 127).to(torch.int8)
del y
out = torch.empty((N // 2) * maxf, dtype=torch.int8, device="cuda")
shapes = [(0, 0, 1), (128, 6, 1), (64, 7, 4), (32, 8, 4), (16, 9, 1), (8, 10, 1)]
sessions = {}
for f in frames:
    row = []
    for fpw, rl, team in shapes:
        if fpw and f % (fpw * team):
            row.append("    -  ")
            continue
        s = sessions.get((fpw, rl, team))
        if s is None:
            s = sessions[(fpw, rl, team)] = polar_amd.Session(0)
            s.configure(fpw, rl, 0, team)
        dec = polar_amd.PolarDecoder(s)
        llr = flat[: N * f].view(N, f)
        o = out[: (N // 2) * f].view(N // 2, f)
        ms = []
        for _ in range(3):
            dec(o, llr, prog)
            ms.append(s.last_decode_ms())
        row.append("%7.2f" % min(ms))
        if not fpw:
            row.append("(%d,%d,%d)" % s.last_decode_shape())
    print("M %d frames %7d: default %s %-10s | (128,6) %s | (64,7)t %s | (32,8)t %s | (16,9) %s | (8,10) %s" % (a.M, f, row[0], row[1], *row[2:]), flush=True)
