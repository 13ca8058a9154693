#!/usr/bin/env python3
"""A/B inside ONE process (one physical placement of the buffers): kernel time of the N = 2^M decode
for several frame counts (row strides) and arena paddings, interleaved and repeated. Consecutive
processes on one box differ by ~5 % for the same call (profiles/r3_process_to_process.txt), so
comparisons across processes say little.
  python3 tools/ab_inproc.py --frames 1048576,1050624 --pads 0,1056 --rounds 3"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=16)
ap.add_argument("--frames", default="1048576")
ap.add_argument("--pads", default="0", help="POLAR_HIP_ARENA_PAD values (dwords)")
ap.add_argument("--hot", default="", help="POLAR_HIP_HOT_LEVELS values (needs a build that reads it per launch)")
ap.add_argument("--strides", default="0", help="extra bytes per row (POLAR_HIP_ROW_STRIDE = frames + this), multiples of 16")
ap.add_argument("--side", default="both", choices=["both", "in", "out"], help="which tensor gets the row stride")
ap.add_argument("--skews", default="", help="POLAR_HIP_SCRATCH_SKEW values (bytes)")
ap.add_argument("--hot-skews", default="", help="POLAR_HIP_HOT_SKEW values (bytes)")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--snr", type=float, default=-1.0)
a = ap.parse_args()
import torch
import polar_amd

N = 1 << a.M
frames = [int(x) for x in a.frames.split(",")]
pads = [int(x) for x in a.pads.split(",")]
fro = np.zeros(N, np.uint8)
polar_amd.PolarCodeConst0()(fro, a.M, N // 2, polar_amd.testbench_probability())
prog = np.zeros(2 * N + 1, np.uint8)
n = polar_amd.PolarCompiler()(prog, fro, a.M)
prog = prog[:n].copy()
sess = polar_amd.Session(0)
sigma = float(np.sqrt(1.0 / (2.0 * 10.0 ** (a.snr / 10.0))))
g = torch.Generator(device="cuda").manual_seed(1)
strides = [int(x) for x in a.strides.split(",")]
maxf = max(frames) + max(strides)
flat = torch.empty(N * maxf, dtype=torch.int8, device="cuda")
step = 1 << 28
for r in range(0, N * maxf, step):
    m = min(step, N * maxf - r)
    y = 1.0 + sigma * torch.randn(m, generator=g, device="cuda")
    flat[r:r + m] = torch.clamp(torch.round(y * (2.0 / sigma**2)), -128, 127).to(torch.int8)
out = torch.empty((N // 2) * maxf, dtype=torch.int8, device="cuda")
dec = polar_amd.PolarDecoder(sess)
os.environ["POLAR_HIP_ARENA_PAD"] = str(max(pads))  # the largest arena first: later launches fit, nothing is re-allocated
skews = [int(x) for x in a.skews.split(",")] if a.skews else [None]
hskews = [int(x) for x in a.hot_skews.split(",")] if a.hot_skews else [None]
if skews[0] is not None:
    os.environ["POLAR_HIP_SCRATCH_SKEW"] = "0"
if hskews[0] is not None:
    os.environ["POLAR_HIP_HOT_SKEW"] = "0"
f0 = max(frames)
dec(out[: (N // 2) * f0].view(N // 2, f0), flat[: N * f0].view(N, f0), prog)
for rnd in range(a.rounds):
    for f in frames:
      for extra in strides:
        knob = {"both": "POLAR_HIP_ROW_STRIDE", "in": "POLAR_HIP_ROW_STRIDE_IN", "out": "POLAR_HIP_ROW_STRIDE_OUT"}[a.side]
        if extra:
            os.environ[knob] = str(f + extra)
        else:
            os.environ.pop(knob, None)
        for pad in pads:
          for sk in skews:
           for hsk in hskews:
            os.environ["POLAR_HIP_ARENA_PAD"] = str(pad)
            if sk is not None:
                os.environ["POLAR_HIP_SCRATCH_SKEW"] = str(sk)
            if hsk is not None:
                os.environ["POLAR_HIP_HOT_SKEW"] = str(hsk)
            llr = flat[: N * f].view(N, f)
            o = out[: (N // 2) * f].view(N // 2, f)
            ms = []
            for _ in range(2):
                dec(o, llr, prog)
                ms.append(sess.last_decode_ms())
            best = min(ms)
            print("round %d frames %d row +%d pad %d skew %s hot skew %s: kernel_ms %.2f  %.1f Gbit/s info  (%.3f us per frame)" % (rnd, f, extra, pad, sk, hsk, best, f * (N // 2) / best / 1e6, best * 1e3 / f), flush=True)
