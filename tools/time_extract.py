#!/usr/bin/env python3
"""Time the systematic post-processing (re-encode + gather) on a resident batch."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=16)
ap.add_argument("--frames", type=int, default=1 << 20)
a = ap.parse_args()
import torch, polar_amd
N = 1 << a.M
fro = np.zeros(N, np.uint8)
polar_amd.PolarCodeConst0()(fro, a.M, N // 2, polar_amd.testbench_probability())
sess = polar_amd.Session(0)
dec = torch.ones((N // 2, a.frames), dtype=torch.int8, device="cuda")
enc = polar_amd.PolarEncoder(a.M, sess)
cols = (a.frames + 31) // 32
sign = torch.zeros((N // 2, 4 * cols), dtype=torch.uint8, device="cuda")
for name, fn in (("int8 in place", lambda: enc.extract_systematic(dec, fro)), ("packed sign plane", lambda: enc.extract_systematic_packed(sign, None, dec, fro))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print("extract %s: M %d frames %d: %.2f ms" % (name, a.M, a.frames, ms), flush=True)
