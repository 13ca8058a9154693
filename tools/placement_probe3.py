#!/usr/bin/env python3
"""One process, one handle, the caller's tensors kept: the decode timed after the handle's scratch arenas and /
or its hot region were freed and allocated anew (POLAR_HIP_REALLOC, experiments only)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import polar_amd

M, frames = 16, 1 << 20
N = 1 << M
fro = np.zeros(N, np.uint8)
polar_amd.PolarCodeConst0()(fro, M, N // 2, polar_amd.testbench_probability())
prog = np.zeros(2 * N + 1, np.uint8)
n = polar_amd.PolarCompiler()(prog, fro, M)
prog = prog[:n].copy()
sigma = float(np.sqrt(1.0 / (2.0 * 10.0 ** (-1.0 / 10.0))))
g = torch.Generator(device="cuda").manual_seed(1)
llr = torch.empty((N, frames), dtype=torch.int8, device="cuda")
for r in range(0, N, 256):
    y = 1.0 + sigma * torch.randn((256, frames), generator=g, device="cuda")
    llr[r:r + 256] = torch.clamp(torch.round(y * (2.0 / sigma**2)), -128, 127).to(torch.int8)
del y
torch.cuda.empty_cache()
out = torch.empty((N // 2, frames), dtype=torch.int8, device="cuda")
sess = polar_amd.Session(0)
dec = polar_amd.PolarDecoder(sess)


def run(label, realloc=None):
    if realloc:
        os.environ["POLAR_HIP_REALLOC"] = realloc
        os.environ["POLAR_HIP_DEBUG"] = "1"
        dec(out, llr, prog)
        os.environ.pop("POLAR_HIP_REALLOC")
        os.environ.pop("POLAR_HIP_DEBUG")
    ms = []
    for _ in range(3):
        dec(out, llr, prog)
        ms.append(sess.last_decode_ms())
    print("%s: %.2f ms" % (label, min(ms)), flush=True)


os.environ["POLAR_HIP_DEBUG"] = "1"
dec(out, llr, prog)
os.environ.pop("POLAR_HIP_DEBUG")
run("first allocation")
for label, what in (("hot region anew", "hot"), ("hot region anew", "hot"), ("arenas anew", "scratch"), ("arenas anew", "scratch"), ("hot region anew", "hot"), ("both anew", "both"), ("arenas anew", "scratch")):
    run(label, what)
