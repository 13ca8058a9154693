#!/usr/bin/env python3
"""HBM ceiling for mixed read/write streams on this box: device-to-device copy and a 3:2
read:write mix (torch kernels), to put the decode kernel's 5.3 TB/s in context."""
import time
import torch

n = 1 << 32  # 4 GiB per buffer
a = torch.empty(n, dtype=torch.uint8, device="cuda").random_()
b = torch.empty_like(a)
c = torch.empty_like(a)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


t = timed(lambda: b.copy_(a))
print("copy      : %.2f TB/s (read + write)" % (2 * n / t / 1e12))
ai, bi, ci = a.view(torch.int32), b.view(torch.int32), c.view(torch.int32)
t = timed(lambda: torch.add(ai, bi, out=ci))
print("add 2r:1w : %.2f TB/s" % (3 * n / t / 1e12))
t = timed(lambda: ai.sum())
print("read only : %.2f TB/s" % (n / t / 1e12))
t = timed(lambda: b.fill_(1))
print("write only: %.2f TB/s" % (n / t / 1e12))
