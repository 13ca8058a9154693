#!/usr/bin/env python3
"""Generate tests/golden/polar_golden.npz + polar_golden.json from the GENUINE reference.

Run in the build container only (needs oracle/_ref/libpolar_ref.so, which oracle/Makefile
compiles from the unmodified headers under /root/reference):

    make -C oracle && python3 tests/golden/make_golden.py

Every expected output below comes from the reference build (`Ref`), never from our own
restatement. Inputs are produced by the deterministic integer generator in
tests/golden/inputs.py (splitmix64), so digest-only vectors can be regenerated anywhere.
The files written are data only: inputs, expected outputs, digests.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from oracle_lib import Ref, fnv1a64  # noqa: E402
from inputs import corner_frames, family_llr, pm1_message  # noqa: E402


def main():
    ref = Ref()
    arrays = {}
    meta = {"generator": "tests/golden/make_golden.py", "reference": "xdsopl/polar @ /root/reference (g++ -O2 -ffast-math -mavx2, SIMD<int8_t,%d>)" % ref.width,
            "fnv": "FNV-1a-64 over raw bytes, offset 0xcbf29ce484222325, prime 0x100000001b3"}

    # (1) code construction + compiler
    codes = {}
    for M in list(range(1, 17)) + [20]:
        N = 1 << M
        entry = {}
        fro = ref.const0(M, N // 2)  # testbench.cc:74-88 probability
        entry["const0_frozen_fnv"] = "%016x" % fnv1a64(fro)
        if M >= 2:  # (2,1) -> frozen 1,1,1,0? level-1 corner handled below
            prog = ref.compile(fro, M)
            entry["const0_prog_len"] = int(len(prog))
            entry["const0_prog_fnv"] = "%016x" % fnv1a64(prog)
        if M <= 10:
            arrays["const0_frozen_m%d" % M] = np.packbits(fro)
            if M >= 2:
                arrays["const0_prog_m%d" % M] = prog
        for tag, th in (("half", 0.5), ("eps", float(np.finfo(np.float32).eps))):
            fz, K = ref.freezer(M, 0.5, th)
            entry["freezer_%s_K" % tag] = int(K)
            entry["freezer_%s_frozen_fnv" % tag] = "%016x" % fnv1a64(fz)
            if 0 < K < N:
                p = ref.compile(fz, M)
                entry["freezer_%s_prog_len" % tag] = int(len(p))
                entry["freezer_%s_prog_fnv" % tag] = "%016x" % fnv1a64(p)
                if M <= 10 and tag == "half":
                    arrays["freezer_half_prog_m%d" % M] = p
            if M <= 10 and tag == "half":
                arrays["freezer_half_frozen_m%d" % M] = np.packbits(fz)
        codes[str(M)] = entry
    meta["codes"] = codes
    # the compiler example of SURVEY 8(a15)
    ex = np.array([1, 1, 1, 0, 1, 0, 0, 0], np.uint8)
    arrays["compile_example_prog"] = ref.compile(ex, 3)

    # (2) decoder KATs, full vectors for small codes
    kats = []
    for M in (3, 5, 8, 10):
        N = 1 << M
        for code in ("const0", "freezer"):
            if code == "const0":
                fro = ref.const0(M, N // 2)
                K = N // 2
            else:
                fro, K = ref.freezer(M, 0.5, 0.5)
            prog = ref.compile(fro, M)
            msg = pm1_message(1000 + M, K, 8)
            cw = ref.encode(msg, fro, M, 0)
            for fam in ("awgn-3", "awgn-1", "awgn+1", "uniform", "small", "corner"):
                if fam == "corner":
                    llr = corner_frames(N)
                else:
                    llr = family_llr(fam, 7 * M + len(fam), cw)
                out = ref.decode(llr, prog, K)
                name = "dec_m%d_%s_%s" % (M, code, fam.replace("-", "n").replace("+", "p"))
                arrays[name + "_out"] = out
                kats.append({"name": name, "M": M, "code": code, "family": fam, "frames": int(llr.shape[1]), "msg_seed": 1000 + M,
                             "llr_seed": 7 * M + len(fam), "in_fnv": "%016x" % fnv1a64(llr), "out_fnv": "%016x" % fnv1a64(out)})
    meta["decoder_kats"] = kats

    # digest-only decoder KATs for the big codes
    big = []
    for M, frames in ((12, 64), (14, 32), (16, 32), (20, 32)):
        N = 1 << M
        fro = ref.const0(M, N // 2)
        prog = ref.compile(fro, M)
        msg = pm1_message(2000 + M, N // 2, frames)
        cw = ref.encode(msg, fro, M, 1)
        for fam in ("awgn-1", "uniform", "small"):
            if M == 20 and fam == "small":
                continue
            llr = family_llr(fam, 11 * M + len(fam), cw)
            out = ref.decode(llr, prog, N // 2)
            big.append({"M": M, "frames": frames, "family": fam, "msg_seed": 2000 + M, "llr_seed": 11 * M + len(fam), "systematic_cw": 1,
                        "in_fnv": "%016x" % fnv1a64(llr), "out_fnv": "%016x" % fnv1a64(out),
                        "zeros_out": int((out == 0).sum()), "errors_vs_u": None})
    meta["decoder_digests"] = big

    # (3) per-op tables (polar_helper.hh:113-173)
    v = np.arange(-128, 128)
    arrays["op_qabs"] = np.array([ref.op("qabs", int(a)) for a in v], np.int8)
    arrays["op_signum"] = np.array([ref.op("signum", int(a)) for a in v], np.int8)
    arrays["op_decide"] = np.array([ref.op("decide", int(a)) for a in v], np.int8)
    prod = np.array([[ref.op("prod", int(a), int(b)) for b in v] for a in v], np.int8)
    qadd = np.array([[ref.op("qadd", int(a), int(b)) for b in v] for a in v], np.int8)
    madd = np.array([[[ref.op("madd", h, int(a), int(b)) for b in v] for a in v] for h in (-1, 0, 1)], np.int8)
    qmul = np.array([[ref.op("qmul", a, b) for b in (-1, 0, 1)] for a in (-1, 0, 1)], np.int8)
    arrays["op_qmul_ternary"] = qmul
    meta["op_digests"] = {"prod": "%016x" % fnv1a64(prod), "qadd": "%016x" % fnv1a64(qadd), "madd": "%016x" % fnv1a64(madd)}
    meta["quant"] = {repr(x): int(ref.L.ref_quant(x)) for x in (-1e9, -128.5, -128.49, -127.5, -2.5, -1.5, -0.5, -0.49, 0.0, 0.5, 0.51, 1.5, 2.5, 126.5, 127.49, 127.5, 1e9)}

    # (4) encoder KATs + systematic round trip (testbench.cc:143-146, 177-183)
    enc = []
    for M in (1, 2, 3, 5, 8, 10):
        N = 1 << M
        fro = ref.const0(M, N // 2)
        msg = pm1_message(3000 + M, N // 2, 5)
        for sysf in (0, 1):
            cw = ref.encode(msg, fro, M, sysf)
            arrays["enc_m%d_sys%d" % (M, sysf)] = cw
            enc.append({"M": M, "systematic": sysf, "frames": 5, "msg_seed": 3000 + M, "cw_fnv": "%016x" % fnv1a64(cw)})
    for M in (12, 16):
        N = 1 << M
        fro = ref.const0(M, N // 2)
        msg = pm1_message(3000 + M, N // 2, 3)
        for sysf in (0, 1):
            cw = ref.encode(msg, fro, M, sysf)
            enc.append({"M": M, "systematic": sysf, "frames": 3, "msg_seed": 3000 + M, "cw_fnv": "%016x" % fnv1a64(cw)})
    # encoder on ternary input (zeros = erasures) as the post-decode re-encode sees it
    M = 8
    fro = ref.const0(M, 128)
    tern = pm1_message(3100, 128, 6)
    tern[::7, :] = 0
    arrays["enc_ternary_in"] = tern
    arrays["enc_ternary_out"] = ref.encode(tern, fro, M, 0)
    meta["encoder_kats"] = enc

    np.savez_compressed(os.path.join(HERE, "polar_golden.npz"), **arrays)
    with open(os.path.join(HERE, "polar_golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", len(arrays), "arrays")


if __name__ == "__main__":
    main()
