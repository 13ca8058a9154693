#!/usr/bin/env python3
"""Randomised soak of the relay launches and pitched rows on the GPU box: random frozen sets at N = 2^9 .. 2^13,
every kernel shape, batches of 1.2 .. 5 sweeps (so that the library sends them as several launches), random row
pitches of input and output; the decisions must equal those of the single launch on contiguous buffers bit for
bit, and the oracle's on a sample of columns. Test infrastructure (the oracle is the checker).
    python tools/soak_relay.py --seconds 120"""
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
a = ap.parse_args()
import torch
import polar_amd
from oracle_lib import Oracle

oracle = Oracle()
sess = polar_amd.Session(0)
dec = polar_amd.PolarDecoder(sess)
variants = [(4, 11, 1), (8, 10, 1), (16, 9, 1), (32, 8, 1), (64, 7, 1), (128, 6, 1), (32, 8, 4), (64, 7, 4), (0, 0, 1)]
slots = {4: 2048, 8: 2048, 16: 2048, 32: 2048, 64: 2048, 128: 2048}
rng = np.random.default_rng(a.seed)
g = torch.Generator(device="cuda").manual_seed(a.seed)
t0 = time.time()
cases = relayed = 0
while time.time() - t0 < a.seconds:
    M = int(rng.integers(9, 14))
    N = 1 << M
    b = 1 << int(rng.integers(0, 6))
    fr = np.repeat((rng.random(N // b) < rng.random()).astype(np.uint8), b)
    for i in range(0, N, 2):
        if fr[i] == 0 and fr[i + 1] == 1:
            fr[i], fr[i + 1] = 1, 0
    try:
        prog = oracle.compile(fr, M)
    except ValueError:
        continue
    K = int((fr == 0).sum())
    if K == 0:
        continue
    fpw, rl, team = variants[int(rng.integers(0, len(variants)))]
    unit = (fpw or 128) * team
    units = int(rng.uniform(1.2, 5.0) * 2048 / team)
    frames = min(units * unit, ((1 << 31) // N) // unit * unit)
    sess.configure(fpw, rl, 0, team)
    cp = frames + (0, 16, 4352, 16 * int(rng.integers(1, 300)))[int(rng.integers(0, 4))]
    mp = frames + (0, 16, 4352, 16 * int(rng.integers(1, 300)))[int(rng.integers(0, 4))]
    lo, hi = ((-128, 128), (-12, 20))[int(rng.integers(0, 2))]
    wide_in = torch.randint(lo, hi, (N, cp), dtype=torch.int8, device="cuda", generator=g)
    llr = wide_in[:, :frames]
    flat = llr.contiguous()
    one = torch.empty((K, frames), dtype=torch.int8, device="cuda")
    os.environ["POLAR_HIP_RELAY"] = "1"
    dec(one, flat, prog)
    os.environ["POLAR_HIP_RELAY"] = "2"  # relay launches whatever the size of the batch
    wide_out = torch.full((K, mp), 91, dtype=torch.int8, device="cuda")
    out = wide_out[:, :frames]
    dec(out, llr, prog)
    os.environ.pop("POLAR_HIP_RELAY", None)
    launches = sess.last_decode_launches()
    relayed += launches > 1
    cols = torch.from_numpy(np.unique(np.concatenate([np.arange(0, 32), rng.integers(0, frames, 32), np.arange(frames - 32, frames)]))).cuda()
    exp = oracle.decode(np.ascontiguousarray(flat[:, cols].cpu().numpy()), prog)
    ok = bool((out == one).all()) and bool((wide_out[:, frames:] == 91).all()) and np.array_equal(out[:, cols].cpu().numpy(), exp)
    if not ok:
        print("MISMATCH M %d frames %d variant (%d, %d, team %d) pitches %d %d launches %d seed %d" % (M, frames, fpw, rl, team, cp, mp, launches, a.seed), flush=True)
        np.save("gpurun_out/soak_relay_fail_frozen.npy", fr)
        sys.exit(1)
    cases += 1
    del wide_in, wide_out, one, flat
    if cases % 20 == 0:
        print("%d cases ok, %d of them sent as several launches (%.0f s)" % (cases, relayed, time.time() - t0), flush=True)
print("relay soak ok: %d cases, %d of them sent as several launches" % (cases, relayed))
