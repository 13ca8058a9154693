#!/usr/bin/env python3
"""Where do relay launches pay? One launch (POLAR_HIP_RELAY=1) against relay launches (=2) in one process, for several
code lengths and batch sizes, LLR rows at the suggested pitch."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import polar_amd

os.environ["POLAR_HIP_PLACEMENT"] = "0"
sess = polar_amd.Session(0)
dec = polar_amd.PolarDecoder(sess)
sigma = float(np.sqrt(1.0 / (2.0 * 10.0 ** (-1.0 / 10.0))))
for M, frames in ((16, 1 << 19), (16, 3 << 18), (16, 1 << 20), (16, 5 << 18), (16, 1 << 21), (15, 1 << 20), (15, 1 << 21), (14, 1 << 21), (12, 1 << 22)):
    N = 1 << M
    fro = np.zeros(N, np.uint8)
    polar_amd.PolarCodeConst0()(fro, M, N // 2, polar_amd.testbench_probability())
    prog = np.zeros(2 * N + 1, np.uint8)
    n = polar_amd.PolarCompiler()(prog, fro, M)
    prog = prog[:n].copy()
    g = torch.Generator(device="cuda").manual_seed(M)
    llr = polar_amd.pitched_empty(N, frames)
    rows = max(1, (1 << 28) // frames)
    for r in range(0, N, rows):
        y = 1.0 + sigma * torch.randn((min(rows, N - r), frames), generator=g, device="cuda")
        llr[r:r + rows] = torch.clamp(torch.round(y * (2.0 / sigma**2)), -128, 127).to(torch.int8)
    del y
    out = torch.empty((N // 2, frames), dtype=torch.int8, device="cuda")
    res = {}
    for rnd in range(3):
        for mode in ("1", "2"):
            os.environ["POLAR_HIP_RELAY"] = mode
            dec(out, llr, prog)
            dec(out, llr, prog)
            res.setdefault(mode, []).append(sess.last_decode_ms())
            launches = sess.last_decode_launches()
    one, relay = min(res["1"]), min(res["2"])
    print("N=2^%d frames %d (2^%d LLRs) shape %s: one launch %.2f ms, relay (%d launches) %.2f ms: %+.1f %%" % (
        M, frames, int(np.log2(N * frames)), sess.last_decode_shape(), one, launches, relay, 100 * (relay / one - 1)), flush=True)
    del llr, out
    torch.cuda.empty_cache()
