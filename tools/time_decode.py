#!/usr/bin/env python3
"""Time the decode kernel for any code size on the all-(+1) codeword + AWGN (the decoder is
symmetric, so this is representative). Prints frames/s and kernel ms.
  python3 tools/time_decode.py --M 16 --frames 32768 --fpw 32 --lds-level 7"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=16)
ap.add_argument("--frames", type=int, default=32768)
ap.add_argument("--sweeps", type=float, default=0, help="set frames to this many full sweeps of the resident wavefronts")
ap.add_argument("--fpw", type=int, default=0)
ap.add_argument("--lds-level", type=int, default=0)
ap.add_argument("--waves", type=int, default=0)
ap.add_argument("--team", type=int, default=1)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--snr", type=float, default=-1.0)
ap.add_argument("--block-frozen", type=int, default=0, help="experiment: make the frozen set constant over 2^B-blocks (majority vote), so that no node below level B is left")
a = ap.parse_args()
import torch
import polar_amd

N = 1 << a.M
if a.sweeps:
    q = a.fpw // 4
    lds = 4 * ((q << (a.lds_level + 1)) * 5 // 4)
    per_cu = min(16, 163840 // lds)
    a.frames = int(a.sweeps * per_cu * 256 * a.fpw)
fro = np.zeros(N, np.uint8)
polar_amd.PolarCodeConst0()(fro, a.M, N // 2, polar_amd.testbench_probability())
if a.block_frozen:
    b = 1 << a.block_frozen
    fro = np.repeat((fro.reshape(-1, b).mean(axis=1) > 0.5).astype(np.uint8), b)
prog = np.zeros(2 * N + 1, np.uint8)
n = polar_amd.PolarCompiler()(prog, fro, a.M)
prog = prog[:n].copy()
sess = polar_amd.Session(0)
sess.configure(a.fpw, a.lds_level, a.waves, a.team)
sigma = float(np.sqrt(1.0 / (2.0 * 10.0 ** (a.snr / 10.0))))
g = torch.Generator(device="cuda").manual_seed(1)
llr = torch.empty((N, a.frames), dtype=torch.int8, device="cuda")
rows = max(1, min(N, (1 << 28) // a.frames))
for r in range(0, N, rows):
    y = 1.0 + sigma * torch.randn((min(rows, N - r), a.frames), generator=g, device="cuda")
    llr[r:r + rows] = torch.clamp(torch.round(y * (2.0 / sigma**2)), -128, 127).to(torch.int8)
Kc = int((fro == 0).sum())
out = torch.empty((max(Kc, 1), a.frames), dtype=torch.int8, device="cuda")
dec = polar_amd.PolarDecoder(sess)
ms = []
for _ in range(a.reps):
    import time
    t0 = time.perf_counter()
    dec(out, llr, prog)
    wall = (time.perf_counter() - t0) * 1e3
    ms.append(sess.last_decode_ms())
    print("  call wall %.2f ms, kernel %.2f ms" % (wall, ms[-1]), flush=True)
best = min(ms)
bad = 0
for r in range(0, min(Kc, N // 2), 1024):
    bad += int((out[r:min(r + 1024, Kc)] != 1).sum())
err = bad / float(max(1, min(Kc, N // 2) * a.frames))
print("M %d frames %d fpw %d rl %d waves %d: kernel_ms %.2f  %.3f Mframes/s  %.1f Gbit/s info  frac %.4f  ber %.2e" % (
    a.M, a.frames, a.fpw, a.lds_level, a.waves, best, a.frames / best / 1e3, a.frames * (N // 2) / best / 1e6,
    a.frames * 1.5 * N / (best * 1e-3) / 8e12, err), flush=True)
