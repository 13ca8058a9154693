"""The plain-C oracle (oracle/polar_oracle.c) against the committed golden vectors, which were
produced by the genuine reference build (tests/golden/make_golden.py). CPU only."""
import numpy as np
import pytest

from inputs import corner_frames, family_llr, pm1_message
from oracle_lib import fnv1a64, program_rows


def _code(oracle, M, code):
    N = 1 << M
    if code == "const0":
        fro, amb = oracle.const0(M, N // 2)
        assert not amb
        return fro, N // 2
    return oracle.freezer(M, 0.5, 0.5)


def test_per_op_tables(oracle, golden):
    meta, arr = golden
    v = np.arange(-128, 128)
    L = oracle.L
    assert [L.oracle_qabs(int(a)) for a in v] == arr["op_qabs"].tolist()
    assert [L.oracle_signum(int(a)) for a in v] == arr["op_signum"].tolist()
    assert [L.oracle_decide(int(a)) for a in v] == arr["op_decide"].tolist()
    prod = np.array([[L.oracle_prod(int(a), int(b)) for b in v] for a in v], np.int8)
    qadd = np.array([[L.oracle_qadd(int(a), int(b)) for b in v] for a in v], np.int8)
    madd = np.array([[[L.oracle_madd(h, int(a), int(b)) for b in v] for a in v] for h in (-1, 0, 1)], np.int8)
    qmul = np.array([[L.oracle_qmul(a, b) for b in (-1, 0, 1)] for a in (-1, 0, 1)], np.int8)
    assert "%016x" % fnv1a64(prod) == meta["op_digests"]["prod"]
    assert "%016x" % fnv1a64(qadd) == meta["op_digests"]["qadd"]
    assert "%016x" % fnv1a64(madd) == meta["op_digests"]["madd"]
    assert (qmul == arr["op_qmul_ternary"]).all()
    # the corner the scalar reference helper gets wrong (hard part 1): a = -128, h = -1
    assert L.oracle_madd(-1, -128, 0) == 127 and L.oracle_madd(-1, -128, 5) == 127 and L.oracle_madd(1, -128, -128) == -128
    for x, q in meta["quant"].items():
        assert L.oracle_quant(float(x)) == q


def test_code_construction_and_compiler(oracle, golden):
    meta, arr = golden
    for M_s, e in meta["codes"].items():
        M = int(M_s)
        N = 1 << M
        fro, amb = oracle.const0(M, N // 2)
        assert not amb
        assert "%016x" % fnv1a64(fro) == e["const0_frozen_fnv"], M
        if M <= 10:
            assert (np.packbits(fro) == arr["const0_frozen_m%d" % M]).all()
        if "const0_prog_len" in e:
            prog = oracle.compile(fro, M)
            assert len(prog) == e["const0_prog_len"] and "%016x" % fnv1a64(prog) == e["const0_prog_fnv"], M
            assert program_rows(prog) == N // 2
        for tag, th in (("half", 0.5), ("eps", float(np.finfo(np.float32).eps))):
            fz, K = oracle.freezer(M, 0.5, th)
            assert K == e["freezer_%s_K" % tag]
            assert "%016x" % fnv1a64(fz) == e["freezer_%s_frozen_fnv" % tag]
            if "freezer_%s_prog_len" % tag in e:
                p = oracle.compile(fz, M)
                assert len(p) == e["freezer_%s_prog_len" % tag] and "%016x" % fnv1a64(p) == e["freezer_%s_prog_fnv" % tag]
    assert oracle.compile(np.array([1, 1, 1, 0, 1, 0, 0, 0], np.uint8), 3).tolist() == arr["compile_example_prog"].tolist() == [3, 0, 5, 1, 6, 2, 255]
    # program lengths / K recorded by the survey's own probe of the reference
    assert [meta["codes"][str(m)]["const0_prog_len"] for m in (10, 11, 12, 13, 14, 15, 16, 20)] == [255, 449, 763, 1375, 2457, 4547, 8419, 114903]
    assert [meta["codes"][str(m)]["freezer_half_prog_len"] for m in range(10, 17)] == [219, 371, 619, 1031, 1727, 2855, 4783]
    assert [meta["codes"][str(m)]["freezer_eps_K"] for m in range(10, 17)] == [241, 563, 1271, 2782, 5994, 12713, 26646]
    with pytest.raises(ValueError):  # [info, frozen] pair: reference asserts level > 0
        oracle.compile(np.array([0, 1], np.uint8), 1)


def test_decoder_kats(oracle, golden):
    meta, arr = golden
    for k in meta["decoder_kats"]:
        M = k["M"]
        N = 1 << M
        fro, K = _code(oracle, M, k["code"])
        prog = oracle.compile(fro, M)
        msg = pm1_message(k["msg_seed"], K, 8)
        cw = oracle.encode(msg, fro, M, 0)
        llr = corner_frames(N) if k["family"] == "corner" else family_llr(k["family"], k["llr_seed"], cw)
        assert "%016x" % fnv1a64(llr) == k["in_fnv"], k["name"]
        out = oracle.decode(llr, prog)
        assert out.shape == arr[k["name"] + "_out"].shape and (out == arr[k["name"] + "_out"]).all(), k["name"]


@pytest.mark.parametrize("M", [12, 14, 16])
def test_decoder_digests(oracle, golden, M):
    meta, _ = golden
    N = 1 << M
    fro, _ = oracle.const0(M, N // 2)
    prog = oracle.compile(fro, M)
    for k in [k for k in meta["decoder_digests"] if k["M"] == M]:
        msg = pm1_message(k["msg_seed"], N // 2, k["frames"])
        cw = oracle.encode(msg, fro, M, 1)
        llr = family_llr(k["family"], k["llr_seed"], cw)
        assert "%016x" % fnv1a64(llr) == k["in_fnv"]
        out = oracle.decode(llr, prog)
        assert "%016x" % fnv1a64(out) == k["out_fnv"], (M, k["family"])
        assert int((out == 0).sum()) == k["zeros_out"]


def test_encoder_kats(oracle, golden):
    meta, arr = golden
    for k in meta["encoder_kats"]:
        M = k["M"]
        N = 1 << M
        fro, _ = oracle.const0(M, N // 2)
        msg = pm1_message(k["msg_seed"], N // 2, k["frames"])
        cw = oracle.encode(msg, fro, M, k["systematic"])
        assert "%016x" % fnv1a64(cw) == k["cw_fnv"], k
        name = "enc_m%d_sys%d" % (M, k["systematic"])
        if name in arr:
            assert (cw == arr[name]).all()
        if k["systematic"]:  # the bench's assert, testbench.cc:143-146
            assert (cw[fro == 0] == msg).all()
    fro, _ = oracle.const0(8, 128)
    assert (oracle.encode(arr["enc_ternary_in"], fro, 8, 0) == arr["enc_ternary_out"]).all()


def test_systematic_round_trip_and_counters(oracle):
    """testbench.cc:129-192 at a noise-free operating point: decode -> re-encode -> gather."""
    M, F = 10, 24
    N = 1 << M
    fro, _ = oracle.const0(M, N // 2)
    prog = oracle.compile(fro, M)
    msg = pm1_message(77, N // 2, F)
    cw = oracle.encode(msg, fro, M, 1)
    llr = (cw.astype(np.int16) * 20).astype(np.int8)
    dec = oracle.systematic_extract(oracle.decode(llr, prog), fro, M)
    assert (dec == msg).all()
    c = oracle.count(llr, cw, dec, msg)
    assert c.tolist() == [0, 0, 0, 0]
    llr2 = llr.copy()
    llr2[0, :] = 0
    llr2[1, :] = -llr2[1, :]
    c = oracle.count(llr2, cw, dec, msg)
    assert c.tolist() == [F, F, 0, 0]


def test_philox_known_answers(oracle):
    """Philox-4x32-10 known-answer vectors of Random123 (kat_vectors: philox4x32 10)."""
    assert oracle.philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert oracle.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert oracle.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_channel_restatement_properties(oracle):
    """testbench.cc:112-114, 151-165: the channel is a pure function of (seed, row, global frame),
    its noise has the bench's sigma, and quantisation follows polar_helper.hh:194-198."""
    rows, frames = 64, 1000
    cw = np.ones((rows, frames), np.int8)
    a = oracle.channel(cw, 30.0, 7)  # sigma = 0.022: llr = round(2/sigma^2 * y) saturates
    assert (a == 127).all()
    snr = 3.0
    sigma = np.sqrt(1.0 / (2.0 * 10.0 ** (snr / 10.0)))
    llr = oracle.channel(cw, snr, 7).astype(np.float64)
    assert np.abs(llr).max() < 127  # not saturated at this SNR
    y = llr * sigma * sigma / 2.0
    assert abs(y.mean() - 1.0) < 4 * sigma / np.sqrt(y.size) + 0.01
    assert abs(y.std() - sigma) < 0.01
    # sharding: frames [300, 700) generated alone equal the slice of the whole batch
    part = oracle.channel(cw[:, 300:700], snr, 7, first_frame=300)
    assert (part == oracle.channel(cw, snr, 7)[:, 300:700]).all()
    m = oracle.random_message(32, 4000, 9)
    assert set(np.unique(m)) == {-1, 1} and abs(m.mean()) < 0.02
    assert (oracle.random_message(32, 100, 9, first_frame=1001) == oracle.random_message(32, 4000, 9)[:, 1001:1101]).all()
    assert (oracle.random_message(32, 100, 10) != m[:, :100]).any()
