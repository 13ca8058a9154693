#!/usr/bin/env python3
"""How much the decode kernels of a rocprofv3 --kernel-trace run overlapped: for the last `n` decode
launches, the union of their intervals against the sum of their durations, and the busiest moment."""
import csv
import glob
import sys

d, n = sys.argv[1], int(sys.argv[2])
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "sc_decode" in r["Kernel_Name"]]
rows = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))[-n:]
iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")) for r in rows]
total = sum(b - a for a, b, _ in iv)
ev = sorted([(a, 1) for a, _, _ in iv] + [(b, -1) for _, b, _ in iv])
cur = peak = 0
union = 0
last = None
for t, s in ev:
    if cur > 0:
        union += t - last
    cur += s
    peak = max(peak, cur)
    last = t
print("%d decode launches on queues %s: sum of durations %.2f ms, union %.2f ms, span %.2f ms, up to %d at once" % (
    len(iv), sorted(set(q for _, _, q in iv)), total / 1e6, union / 1e6, (max(b for _, b, _ in iv) - min(a for a, _, _ in iv)) / 1e6, peak))
