#!/usr/bin/env python3
"""Time polar_hip_decode with host buffers (POLAR_HOST): pageable numpy arrays and pinned ones."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=16)
ap.add_argument("--frames", type=int, default=1 << 16)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
import torch, polar_amd
N = 1 << a.M
K = N // 2
fro = np.zeros(N, np.uint8)
polar_amd.PolarCodeConst0()(fro, a.M, K, polar_amd.testbench_probability())
prog = np.zeros(2 * N + 1, np.uint8)
n = polar_amd.PolarCompiler()(prog, fro, a.M)
prog = prog[:n].copy()
sess = polar_amd.Session(0)
dec = polar_amd.PolarDecoder(sess)
rng = np.random.default_rng(1)
row = np.clip(np.rint((1.0 + 0.8 * rng.standard_normal(a.frames)) * 3.0), -127, 127).astype(np.int8)
for kind in ("pageable", "pinned"):
    if kind == "pageable":
        llr = np.empty((N, a.frames), np.int8)
        out = np.empty((K, a.frames), np.int8)
    else:
        tl = torch.empty((N, a.frames), dtype=torch.int8).pin_memory()
        to = torch.empty((K, a.frames), dtype=torch.int8).pin_memory()
        llr, out = tl.numpy(), to.numpy()
    llr[:] = row[None, :]
    llr[1::2] = np.roll(row, 7)[None, :]
    for r in range(a.reps + 1):
        t0 = time.perf_counter()
        dec(out, llr, prog)
        dt = time.perf_counter() - t0
        if r:
            print("host path (%s): M %d frames %d: %.1f ms  %.3f Mframes/s  %.1f Gbit/s info  %.1f GB/s over the bus (N in + K out)  kernel %.1f ms  sum %d" % (
                kind, a.M, a.frames, dt * 1e3, a.frames / dt / 1e6, a.frames * K / dt / 1e9, (N + K) * a.frames / dt / 1e9, sess.last_decode_ms(), int(out[:, :64].sum())), flush=True)
