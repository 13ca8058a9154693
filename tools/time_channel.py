#!/usr/bin/env python3
"""Time polar_hip_channel on a resident batch: screened single precision (default) and double only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, polar_amd
sess = polar_amd.Session(0)
rows, frames = 65536, 1 << 16
cw = torch.ones((rows, frames), dtype=torch.int8, device="cuda")
out = torch.empty_like(cw)
for snr in (-1.0, 5.0):
    for mode in ("screened", "double only"):
        if mode == "double only":
            os.environ["POLAR_HIP_CHANNEL_F64"] = "1"
        else:
            os.environ.pop("POLAR_HIP_CHANNEL_F64", None)
        sess.channel(out, cw, snr, 3, 0); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): sess.channel(out, cw, snr, 3, 0)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        print("channel %s, Es/N0 %.1f dB: %d symbols in %.2f ms = %.0f Gsymbol/s  sum %d" % (mode, snr, rows * frames, ms, rows * frames / ms / 1e6, int(out[:64].to(torch.int64).sum().item())), flush=True)
