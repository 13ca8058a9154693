#!/usr/bin/env python3
"""One process, one handle, the caller's tensors kept: the decode timed after the handle's scratch arenas were
freed and allocated anew, plainly (hipMalloc) or as separately allocated pieces of the given size in MiB mapped
into one range, shuffled with an 's' (POLAR_HIP_REALLOC, POLAR_HIP_ARENA_PIECES: experiments only).
  python3 tools/placement_probe4.py plain 2048 2048s 256 256s 32 32s plain"""
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


os.environ["POLAR_HIP_PLACEMENT"] = "0"
dec(out, llr, prog)
run("plain allocation, first")
import time
for pieces in sys.argv[1:]:
    if pieces == "plain":
        os.environ.pop("POLAR_HIP_ARENA_PIECES", None)
    else:
        os.environ["POLAR_HIP_ARENA_PIECES"] = pieces
    t0 = time.perf_counter()
    os.environ["POLAR_HIP_REALLOC"] = "scratch"
    dec(out, llr, prog)
    os.environ.pop("POLAR_HIP_REALLOC")
    dt = time.perf_counter() - t0
    run("arenas anew, pieces %-6s (call with allocation %.2f s)" % (pieces, dt))
