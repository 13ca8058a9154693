#!/usr/bin/env python3
"""Time the float decoder (polar_hip_decode_f32) on a resident batch and compare its decisions
with the int8 decoder's on the same quantised LLRs."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=16)
ap.add_argument("--frames", type=int, default=16384)
ap.add_argument("--snr", type=float, default=-1.0)
a = ap.parse_args()
import torch, polar_amd
N = 1 << a.M
sess = polar_amd.Session(0)
fro, prog, _ = sess.code(a.M, N // 2, polar_amd.testbench_probability())
sigma = float(np.sqrt(1.0 / (2.0 * 10.0 ** (a.snr / 10.0))))
g = torch.Generator(device="cuda").manual_seed(1)
y = 1.0 + sigma * torch.randn((N, a.frames), generator=g, device="cuda")
llr8 = torch.clamp(torch.round(y * (2.0 / sigma**2)), -128, 127).to(torch.int8)
llrf = llr8.to(torch.float32)
dec = polar_amd.PolarDecoder(sess)
o8 = torch.empty((N // 2, a.frames), dtype=torch.int8, device="cuda")
of = torch.empty((N // 2, a.frames), dtype=torch.float32, device="cuda")
dec(o8, llr8, prog); ms8 = sess.last_decode_ms()
dec(of, llrf, prog); dec(of, llrf, prog); msf = sess.last_decode_ms()
agree = float((of.sign().to(torch.int8) == o8).float().mean())
print("M %d frames %d: float %.2f ms (%.1f Gbit/s), int8 %.2f ms (%.1f Gbit/s); decisions equal on %.6f of the symbols; errors float %d int8 %d" % (
    a.M, a.frames, msf, a.frames * N / 2 / msf / 1e6, ms8, a.frames * N / 2 / ms8 / 1e6, agree, int((of != 1).sum()), int((o8 != 1).sum())), flush=True)
