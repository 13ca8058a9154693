#!/usr/bin/env python3
"""Time polar_hip_simulate (one SNR point of the bench loop, everything on the device)."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=16)
ap.add_argument("--frames", type=int, default=1 << 20)
ap.add_argument("--systematic", type=int, default=1)
a = ap.parse_args()
import torch, polar_amd
N = 1 << a.M
fro = np.zeros(N, np.uint8)
polar_amd.PolarCodeConst0()(fro, a.M, N // 2, polar_amd.testbench_probability())
prog = np.zeros(2 * N + 1, np.uint8)
n = polar_amd.PolarCompiler()(prog, fro, a.M)
prog = prog[:n].copy()
sess = polar_amd.Session(0)
for rep in range(3):
    c = np.zeros(6, np.int64)
    t0 = time.perf_counter()
    sess.simulate(c, prog, fro, a.M, a.systematic, -1.0, 7, 0, a.frames)
    dt = time.perf_counter() - t0
    print("simulate M %d systematic %d frames %d: %.1f ms  %.2f Mframes/s  %.1f Gbit/s info end to end  counters %s  (last decode chunk: %s, %.1f ms)" % (
        a.M, a.systematic, a.frames, dt * 1e3, a.frames / dt / 1e6, a.frames * (N // 2) / dt / 1e9, c.tolist(), sess.last_decode_shape(), sess.last_decode_ms()), flush=True)
