#!/usr/bin/env python3
"""Does the tail of one decode launch (the older wavefront of every SIMD finishes 11 % earlier) overlap with the
head of the next when the two run on different hardware queues? The 2^20-frame batch of N = 2^16 as one launch,
and as two halves (frames [0, 2^19) and [2^19, 2^20) of the same tensors, row pitch 2^20: POLAR_HIP_ROW_STRIDE) on
two handles and two streams, for several pairs of torch's pooled streams."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import polar_amd

os.environ["POLAR_HIP_PLACEMENT"] = "0"
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
out = torch.empty((N // 2, frames), dtype=torch.int8, device="cuda")
ref = torch.empty((N // 2, frames), dtype=torch.int8, device="cuda")
whole = polar_amd.Session(0)
dw = polar_amd.PolarDecoder(whole)
for _ in range(3):
    dw(ref, llr, prog)
print("one launch: %.2f ms" % whole.last_decode_ms(), flush=True)
half = frames // 2
os.environ["POLAR_HIP_ROW_STRIDE"] = str(frames)
sa, sb = polar_amd.Session(0), polar_amd.Session(0)
da, db = polar_amd.PolarDecoder(sa), polar_amd.PolarDecoder(sb)


class View:  # the first `half` frames of every row, starting at column c0: same row pitch as the whole tensor
    def __init__(self, t, c0):
        self.t, self.c0 = t, c0
        self.shape = (t.shape[0], half)
        self.ndim = 2
        self.dtype = t.dtype
        self.is_cuda = True
        self.device = t.device

    def data_ptr(self):
        return self.t.data_ptr() + self.c0

    def is_contiguous(self):
        return True


streams = [torch.cuda.Stream() for _ in range(6)]
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for ia, ib in ((0, 0), (0, 1), (0, 2), (0, 3), (1, 2), (0, 4), (0, 5)):
    s0, s1 = streams[ia], streams[ib]
    best = 1e9
    for rep in range(3):
        out.fill_(0)
        torch.cuda.synchronize()
        ev0.record(s0)
        s1.wait_event(ev0)
        da.launch(View(out, 0), View(llr, 0), prog, s0.cuda_stream)
        db.launch(View(out, half), View(llr, half), prog, s1.cuda_stream)
        s0.wait_stream(s1)
        ev1.record(s0)
        torch.cuda.synchronize()
        best = min(best, ev0.elapsed_time(ev1))
    ok = bool((out == ref).all())
    print("two halves on streams %d and %d: %.2f ms, results equal: %s" % (ia, ib, best, ok), flush=True)
