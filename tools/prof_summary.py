#!/usr/bin/env python3
"""Condense rocprofv3 csv output (kernel stats + pmc passes) for the decode kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats", f)
    for row in csv.DictReader(open(f)):
        print("  %-60s calls %s avg_ns %s total_ns %s pct %s" % (row.get("Name", "")[:60], row.get("Calls"), row.get("AverageNs"), row.get("TotalDurationNs"), row.get("Percentage")))
vals = defaultdict(list)
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "sc_decode" in row.get("Kernel_Name", ""):
            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("== counters for sc_decode_kernel (per dispatch, mean over dispatches of the batch-sized launches)")
for k in sorted(vals):
    v = vals[k]
    big = [x for x in v if x >= 0.5 * max(v)] or v
    print("  %-28s %.6g   (n=%d)" % (k, sum(big) / len(big), len(big)))
