#!/usr/bin/env python3
"""Randomised parity soak on the GPU box: random frozen sets (every node type, block-structured and
not), random frame counts and input families, every kernel variant, HIP decode vs the oracle.
Test infrastructure (the oracle is the checker). Prints one line per case set; exits 1 on a mismatch.
    python tools/soak_gpu.py --seconds 120"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=60)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--min-m", type=int, default=3)
ap.add_argument("--max-m", type=int, default=12)
a = ap.parse_args()
import polar_amd
from inputs import family_llr, rand_bytes
from oracle_lib import Oracle

oracle = Oracle()
sess = polar_amd.Session(0)
dec = polar_amd.PolarDecoder(sess)
variants = [(0, 0, 1), (4, 11, 1), (8, 10, 1), (16, 9, 1), (32, 8, 1), (64, 7, 1), (128, 6, 1), (16, 8, 1), (32, 7, 1), (64, 6, 1), (256, 5, 1), (128, 5, 1), (16, 10, 1),
            (32, 8, 4), (64, 7, 4)]  # (frames per wavefront, LDS level, team)
rng = np.random.default_rng(a.seed)
t0 = time.time()
cases = 0
while time.time() - t0 < a.seconds:
    M = int(rng.integers(a.min_m, a.max_m + 1))
    N = 1 << M
    b = 1 << int(rng.integers(0, min(M, 6) + 1))
    fr = np.repeat((rng.random(N // b) < rng.random()).astype(np.uint8), b)
    for i in range(0, N, 2):  # an information bit directly left of a frozen one cannot be compiled
        if fr[i] == 0 and fr[i + 1] == 1:
            fr[i], fr[i + 1] = 1, 0
    try:
        prog = oracle.compile(fr, M)
    except ValueError:
        continue
    K = int((fr == 0).sum())
    frames = int(rng.integers(1, 300)) if rng.random() < 0.8 else int(rng.integers(300, 2500))  # the long ones: several teams, several sweeps
    fam = ("uniform", "small", "awgn-1")[int(rng.integers(0, 3))]
    if fam == "uniform":
        llr = rand_bytes(int(rng.integers(1 << 30)), (N, frames)).view(np.int8)
    else:
        cw = oracle.encode(np.ones((max(K, 1), frames), np.int8)[:K], fr, M, 0)
        llr = family_llr(fam, int(rng.integers(1 << 30)), cw)
    exp = oracle.decode(llr, prog)
    for fpw, rl, team in variants:
        sess.configure(fpw, rl, int(rng.integers(0, 3)) * 8, team)
        out = np.zeros((max(K, 1), frames), np.int8)
        dec(out, np.ascontiguousarray(llr), prog)
        if not (out[:K] == exp).all():
            print("MISMATCH M %d frames %d fam %s variant (%d, %d, team %d) block %d seed %d" % (M, frames, fam, fpw, rl, team, b, a.seed), flush=True)
            np.save("gpurun_out/soak_fail_frozen.npy", fr)
            sys.exit(1)
    cases += 1
    if cases % 50 == 0:
        print("%d frozen sets x %d variants ok (%.0f s)" % (cases, len(variants), time.time() - t0), flush=True)
print("soak ok: %d frozen sets x %d variants" % (cases, len(variants)))
