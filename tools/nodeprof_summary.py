#!/usr/bin/env python3
"""Summarise the per-node cycle dump of the POLAR_PROFILE build (tools/time_decode.py output)."""
import collections
import re
import sys

rows = []
for line in open(sys.argv[1]):
    if line.startswith("M "):
        print(line.strip())
    m = re.match(r'\s+(\S+)\s+L(\d+)\s+calls/wave\s+([\d.]+)\s+cycles/call\s+(\d+)\s+share\s+([\d.]+)%', line)
    if m:
        rows.append((m.group(1), int(m.group(2)), float(m.group(3)), float(m.group(4)), float(m.group(5))))
rl = int(sys.argv[2]) if len(sys.argv) > 2 else 7
bylvl = collections.defaultdict(float)
bykind = collections.defaultdict(float)
tot = 0
for k, l, c, cy, sh in rows:
    bylvl[l] += sh
    bykind[(k, 'hbm' if l > rl else 'lds')] += sh
    tot += c * cy
print('cycles/wave %.1fM' % (tot / 1e6))
print(' by level:', ' '.join('L%d:%.1f' % (l, bylvl[l]) for l in sorted(bylvl)))
print(' by kind:', ' '.join('%s/%s:%.1f' % (k[0], k[1], v) for k, v in sorted(bykind.items(), key=lambda x: -x[1])))
for r in sorted(rows, key=lambda r: -r[4])[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print('%-8s L%-2d calls %7.1f cyc/call %9.0f share %5.2f%%' % r)
