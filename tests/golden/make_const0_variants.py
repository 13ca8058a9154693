#!/usr/bin/env python3
"""Generate tests/golden/const0_variants.json from the GENUINE reference (oracle/_ref, build
container only): PolarCodeConst0 (polar_freezer.hh:49-61) + PolarCompiler at rates 1/4, 1/2, 3/4
and several erasure probabilities (design points, testbench.cc:74-88). Data only: digests.

    make -C oracle && python3 tests/golden/make_const0_variants.py
"""
import json
import math
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle_lib import Oracle, Ref, fnv1a64  # noqa: E402


def design_probability(snr_db):
    """testbench.cc:85-87 with the design SNR as a parameter."""
    return math.exp(-(10.0 ** ((snr_db + 1.59175) / 10.0)))


def main():
    ref, orc = Ref(), Oracle()
    out = []
    ps = [("exp(-1)", math.exp(-1.0)), ("0.3", 0.3), ("0.5", 0.5), ("0.7", 0.7)] + [("design %.1f dB" % s, design_probability(s)) for s in (-4.0, -1.59175, 0.0, 2.0)]
    for M in (6, 8, 10, 12, 14, 16):
        N = 1 << M
        for K in (N // 4, N // 2, 3 * N // 4):
            for name, p in ps:
                fro = ref.const0(M, K, p)
                _, amb = orc.const0(M, K, p)  # metadata: an exact tie at the K boundary makes the reference's pick implementation-defined
                e = {"M": M, "K": K, "p": repr(p), "p_name": name, "frozen_fnv": "%016x" % fnv1a64(fro), "tie_at_boundary": int(amb)}
                try:
                    orc.compile(fro, M)  # the reference aborts (assert level > 0) on an [info, frozen] pair: ask first
                    compilable = True
                except ValueError:
                    compilable = False
                e["compilable"] = int(compilable)
                if compilable:
                    prog = ref.compile(fro, M)
                    e["prog_len"] = int(len(prog))
                    e["prog_fnv"] = "%016x" % fnv1a64(prog)
                out.append(e)
    with open(os.path.join(HERE, "const0_variants.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_const0_variants.py", "cases": out}, f, indent=1, sort_keys=True)
    print("wrote", len(out), "cases,", sum(c["tie_at_boundary"] for c in out), "with a tie at the boundary")


main()
