#!/usr/bin/env python3
"""Model of the HBM bytes per frame the decode kernel moves for a (lds_level) choice, from the
reference byte-code program: which node kinds and levels the scratch traffic comes from."""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import polar_amd  # noqa: E402  (host-side construction and compiler of the library itself)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=16)
    ap.add_argument("--rl", type=int, default=7)
    ap.add_argument("--fuse", type=int, default=3)
    a = ap.parse_args()
    N = 1 << a.M
    fro = np.zeros(N, np.uint8)
    polar_amd.PolarCodeConst0()(fro, a.M, N // 2, polar_amd.testbench_probability())
    buf = np.zeros(2 * N + 2, np.uint8)
    n = polar_amd.PolarCompiler()(buf, fro, a.M)
    prog = [int(x) for x in buf[:n]]
    top, RL = a.M, a.rl
    rd = collections.Counter()
    wr = collections.Counter()
    nops = collections.Counter()
    lvl = prog[0]
    seq = []
    for op in prog[1:]:
        if op == 255:
            break
        if op in (0, 7):
            seq.append((op, lvl))
            lvl -= 1
        elif op == 1:
            seq.append((op, lvl + 1))
        elif op in (2, 8, 9):
            lvl += 1
            seq.append((op, lvl))
        else:
            seq.append((op, lvl))
    def soft_rd(l, key, mult=1.0):
        if l > RL:
            rd[(key, l)] += mult * (1 << l)
    def soft_wr(l, key):
        if l > RL:
            wr[(key, l)] += 1 << l
    def plane_rd(l, key, n):  # n elements at node level l (planes of a level-l node live in HBM if l > RL)
        if l > RL:
            rd[(key + "_pl", l)] += 2 * n / 8
    def plane_wr(l, key, n):
        if l > RL:
            wr[(key + "_pl", l)] += 2 * n / 8
    k = 0
    while k < len(seq):
        op, l = seq[k]
        nops[(op, l)] += 1
        if op in (0, 1, 7):
            # fused chain
            depth = 1
            while a.fuse > depth and k + depth < len(seq) and seq[k + depth][0] in (0, 7) and seq[k + depth][1] == l - depth:
                depth += 1
            soft_rd(l, "chain")
            if op == 1:
                plane_rd(l - 1, "chain", 1 << (l - 1))
            for d in range(1, depth + 1):
                skip = d < depth and seq[k + d][0] == 7 and False
                soft_wr(l - d, "chain")
            k += depth
            continue
        if op in (2, 8):
            h = 1 << (l - 1)
            if l == RL + 1:
                plane_wr(l, "comb", 2 * h)
            elif l > RL + 1:
                plane_rd(l, "comb", 2 * h if op == 2 else h)
                plane_wr(l, "comb", h)
        elif op == 9:
            h = 1 << (l - 1)
            soft_rd(l, "r1comb")
            plane_rd(l - 1, "r1comb", h)
            plane_wr(l, "r1comb", 2 * h)
            if l - 1 > RL:
                wr[("r1comb_tmp", l)] += 2 * h / 8 * 2
                rd[("r1comb_tmp", l)] += 2 * h / 8 * 4
            wr[("out", l)] += h
        else:
            n = 1 << l
            name = {3: "rate0", 4: "rate1", 5: "rep", 6: "spc"}[op]
            if op == 3:
                plane_wr(l, name, n)
            elif op == 5:
                if l > 3:
                    for ll in range(l, 0, -1):
                        soft_rd(ll, name)
                        soft_wr(ll - 1, name)
                plane_wr(l, name, n)
                wr[("out", l)] += 1
            else:
                soft_rd(l, name, 2.0 if op == 6 else 1.0)
                plane_wr(l, name, n)
                if l - 1 > RL:
                    wr[(name + "_tmp", l)] += n / 8 * (6 if op == 6 else 2)
                    rd[(name + "_tmp", l)] += n / 8 * 8
                wr[("out", l)] += n - (op == 6)
        k += 1
    tot_r = sum(rd.values())
    tot_w = sum(wr.values())
    print("M %d RL %d fuse %d: read %.0f KB/frame, write %.0f KB/frame, total %.0f KB (algorithmic %.0f KB)" % (a.M, RL, a.fuse, tot_r / 1024, tot_w / 1024, (tot_r + tot_w) / 1024, 1.5 * N / 1024))
    bykey = collections.Counter()
    for (kk, l), v in rd.items():
        bykey["rd " + kk] += v
    for (kk, l), v in wr.items():
        bykey["wr " + kk] += v
    for kk, v in bykey.most_common():
        print("  %-16s %8.1f KB" % (kk, v / 1024))
    bylvl = collections.Counter()
    for (kk, l), v in list(rd.items()) + list(wr.items()):
        bylvl[l] += v
    print("  by level:", " ".join("L%d:%.0fK" % (l, bylvl[l] / 1024) for l in sorted(bylvl)))
    leaves = collections.Counter()
    for (op, l), c in nops.items():
        leaves[(op, l)] = c
    names = {0: "left", 1: "right", 2: "comb", 3: "rate0", 4: "rate1", 5: "rep", 6: "spc", 7: "r0right", 8: "r0comb", 9: "r1comb"}
    for l in range(top, 0, -1):
        print("  L%-2d %s" % (l, " ".join("%s:%d" % (names[op], c) for (op, ll), c in sorted(nops.items()) if ll == l)))


main()
