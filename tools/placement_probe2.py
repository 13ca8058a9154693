#!/usr/bin/env python3
"""Which buffers' placement matters: the library's scratch (new handle per run, the caller's tensors kept)
or the caller's LLR / message tensors (re-allocated per run, the handle kept)? One process."""
import gc
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


def make_llr():
    g = torch.Generator(device="cuda").manual_seed(1)
    llr = torch.empty((N, frames), dtype=torch.int8, device="cuda")
    rows = 256
    for r in range(0, N, rows):
        y = 1.0 + sigma * torch.randn((rows, frames), generator=g, device="cuda")
        llr[r:r + rows] = torch.clamp(torch.round(y * (2.0 / sigma**2)), -128, 127).to(torch.int8)
    return llr


def time_it(sess, out, llr):
    dec = polar_amd.PolarDecoder(sess)
    ms = []
    for _ in range(3):
        dec(out, llr, prog)
        ms.append(sess.last_decode_ms())
    return min(ms)


llr = make_llr()
out = torch.empty((N // 2, frames), dtype=torch.int8, device="cuda")
for k in range(5):
    hold = torch.empty((k * 3) << 30, dtype=torch.int8, device="cuda") if k else None  # moves the next scratch allocation
    sess = polar_amd.Session(0)
    print("new handle %d (tensors kept): %.2f ms" % (k, time_it(sess, out, llr)), flush=True)
    sess.close()
    del sess, hold
    gc.collect()
    torch.cuda.empty_cache()
sess = polar_amd.Session(0)
print("handle kept from here on: %.2f ms" % time_it(sess, out, llr), flush=True)
for k in range(4):
    del llr, out
    gc.collect()
    torch.cuda.empty_cache()
    hold = torch.empty((k * 5 + 1) << 30, dtype=torch.int8, device="cuda")
    llr = make_llr()
    out = torch.empty((N // 2, frames), dtype=torch.int8, device="cuda")
    del hold
    print("new tensors %d (handle kept): llr 0x%x out 0x%x: %.2f ms" % (k, llr.data_ptr(), out.data_ptr(), time_it(sess, out, llr)), flush=True)
