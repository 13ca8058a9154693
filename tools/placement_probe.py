#!/usr/bin/env python3
"""Does the physical placement of the buffers decide the ~5 % spread between processes? In ONE process
the same N = 2^16, 2^20-frame decode is timed with its buffers allocated behind dummy allocations of
several sizes (so that they land elsewhere in device memory), everything freed in between.
  python3 tools/placement_probe.py --dummies 0,64,128,0"""
import argparse
import gc
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--dummies", default="0,64,128,0", help="GiB allocated before the decode's buffers, one run each")
ap.add_argument("--M", type=int, default=16)
ap.add_argument("--frames", type=int, default=1 << 20)
a = ap.parse_args()
import torch
import polar_amd

N = 1 << a.M
fro = np.zeros(N, np.uint8)
polar_amd.PolarCodeConst0()(fro, a.M, N // 2, polar_amd.testbench_probability())
prog = np.zeros(2 * N + 1, np.uint8)
n = polar_amd.PolarCompiler()(prog, fro, a.M)
prog = prog[:n].copy()
sigma = float(np.sqrt(1.0 / (2.0 * 10.0 ** (-1.0 / 10.0))))
for gib in [int(x) for x in a.dummies.split(",")]:
    dummy = torch.empty(gib << 30, dtype=torch.int8, device="cuda") if gib else None
    sess = polar_amd.Session(0)
    g = torch.Generator(device="cuda").manual_seed(1)
    llr = torch.empty((N, a.frames), dtype=torch.int8, device="cuda")
    rows = max(1, min(N, (1 << 28) // a.frames))
    for r in range(0, N, rows):
        y = 1.0 + sigma * torch.randn((min(rows, N - r), a.frames), generator=g, device="cuda")
        llr[r:r + rows] = torch.clamp(torch.round(y * (2.0 / sigma**2)), -128, 127).to(torch.int8)
    del y
    out = torch.empty((N // 2, a.frames), dtype=torch.int8, device="cuda")
    dec = polar_amd.PolarDecoder(sess)
    ms = []
    for _ in range(4):
        dec(out, llr, prog)
        ms.append(sess.last_decode_ms())
    print("dummy %3d GiB first: llr at 0x%x, out at 0x%x: kernel_ms %s" % (gib, llr.data_ptr(), out.data_ptr(), " ".join("%.2f" % m for m in ms)), flush=True)
    sess.close()
    del llr, out, dec, sess, dummy
    gc.collect()
    torch.cuda.empty_cache()
