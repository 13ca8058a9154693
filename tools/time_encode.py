#!/usr/bin/env python3
"""Time the encoders on a resident batch (POLAR_HIP_ENCODE_BYTES=1 selects the byte-symbol passes)."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=16)
ap.add_argument("--frames", type=int, default=1 << 19)
a = ap.parse_args()
import torch, polar_amd
N = 1 << a.M
K = N // 2
fro = np.zeros(N, np.uint8)
polar_amd.PolarCodeConst0()(fro, a.M, K, polar_amd.testbench_probability())
sess = polar_amd.Session(0)
g = torch.Generator(device="cuda").manual_seed(3)
msg = (1 - 2 * torch.randint(0, 2, (K, a.frames), generator=g, device="cuda", dtype=torch.int8))
cw = torch.empty((N, a.frames), dtype=torch.int8, device="cuda")
path = "byte passes" if os.environ.get("POLAR_HIP_ENCODE_BYTES") else "packed passes"
for name, enc in (("PolarEncoder", polar_amd.PolarEncoder(a.M, sess)), ("PolarSysEnc", polar_amd.PolarSysEnc(a.M, sess))):
    enc(cw, msg, fro); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): enc(cw, msg, fro)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    alg = (N + K) * a.frames
    print("%s (%s): M %d frames %d: %.2f ms  %.1f Gsymbol/s  %.0f GB/s algorithmic (K in + N out) = %.3f of 8 TB/s  fnv %016x" % (
        name, path, a.M, a.frames, ms, N * a.frames / ms / 1e6, alg / ms / 1e6, alg / ms / 1e6 / 8000, int(cw[:, :32].to(torch.int64).sum().item()) & 0xFFFFFFFFFFFFFFFF), flush=True)
