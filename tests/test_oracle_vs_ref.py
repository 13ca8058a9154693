"""The plain-C oracle against the genuine reference build (oracle/_ref/libpolar_ref.so, compiled
from the unmodified headers under /root/reference by oracle/Makefile) on fresh random inputs.
Skipped where that build is not available. CPU only."""
import numpy as np
import pytest

from oracle_lib import awgn_llr, program_rows


def test_ops_exhaustive(oracle, ref):
    L = oracle.L
    for a in range(-128, 128):
        for name in ("qabs", "signum", "decide"):
            assert getattr(L, "oracle_" + name)(a) == ref.op(name, a)
        for b in range(-128, 128):
            for name in ("prod", "qadd", "qmin"):
                assert getattr(L, "oracle_" + name)(a, b) == ref.op(name, a, b), (name, a, b)
            for h in (-1, 0, 1):
                assert L.oracle_madd(h, a, b) == ref.op("madd", h, a, b), (h, a, b)


@pytest.mark.parametrize("M", list(range(1, 15)))
def test_construct_compile_encode_decode(oracle, ref, M):
    rng = np.random.default_rng(100 + M)
    N = 1 << M
    for K in sorted({N // 2, max(1, N // 4), min(N - 1, 3 * N // 4)}):
        f1, amb = oracle.const0(M, K)
        assert not amb and (f1 == ref.const0(M, K)).all()
        p1 = oracle.compile(f1, M)
        assert p1.tolist() == ref.compile(f1, M).tolist()
        assert program_rows(p1) == K
        F = 37 if M <= 11 else 33
        msg = (1 - 2 * rng.integers(0, 2, (K, F))).astype(np.int8)
        for sysf in (0, 1):
            assert (oracle.encode(msg, f1, M, sysf) == ref.encode(msg, f1, M, sysf)).all()
        cw = oracle.encode(msg, f1, M, 1)
        for fam in ("awgn", "uniform", "small"):
            if fam == "awgn":
                llr = awgn_llr(rng, cw, -1.0)
            elif fam == "uniform":
                llr = rng.integers(-128, 128, (N, F)).astype(np.int8)
            else:
                llr = rng.integers(-3, 4, (N, F)).astype(np.int8)
            assert (oracle.decode(llr, p1) == ref.decode(llr, p1, K)).all(), (M, K, fam)
    for th in (0.5, float(np.finfo(np.float32).eps)):
        a, Ka = oracle.freezer(M, 0.5, th)
        b, Kb = ref.freezer(M, 0.5, th)
        assert Ka == Kb and (a == b).all()


def test_random_frozen_sets(oracle, ref):
    """Arbitrary (non polar-ordered) frozen sets reach every opcode, rate0 / rate1 at any level."""
    rng = np.random.default_rng(5)
    seen = set()
    for M in range(1, 9):
        N = 1 << M
        for trial in range(40):
            b = 1 << int(rng.integers(0, M + 1))
            fr = np.repeat((rng.random(N // b) < rng.random()).astype(np.uint8), b)
            for i in range(0, N, 2):  # [info, frozen] cannot be compiled
                if fr[i] == 0 and fr[i + 1] == 1:
                    fr[i], fr[i + 1] = 1, 0
            prog = oracle.compile(fr, M)
            assert prog.tolist() == ref.compile(fr, M).tolist()
            seen.update(prog[1:-1].tolist())
            K = int((fr == 0).sum())
            llr = rng.integers(-128, 128, (N, 33)).astype(np.int8)
            assert (oracle.decode(llr, prog) == ref.decode(llr, prog, K)).all()
    assert seen == set(range(10))


def test_level20(oracle, ref):
    M = 20
    N = 1 << M
    f1, amb = oracle.const0(M, N // 2)
    assert not amb and (f1 == ref.const0(M, N // 2)).all()
    p = oracle.compile(f1, M)
    assert p.tolist() == ref.compile(f1, M).tolist() and len(p) == 114903
    llr = np.random.default_rng(1).integers(-128, 128, (N, 3)).astype(np.int8)
    assert (oracle.decode(llr, p) == ref.decode(llr, p, N // 2)).all()


def test_float_decoder_restatement(oracle, ref):
    """the float path (PolarHelper<SIMD<float,W>>, polar_helper.hh:62-111) restated per
    lane against the reference's own SIMD<float, 8> decoder, bit for bit including the sign of
    zeros: integer-valued LLRs (the quantised channel output, zeros included), fractional LLRs
    and inputs with many exact zeros."""
    import numpy as np

    from inputs import family_llr, pm1_message, rand_bytes

    for M in (3, 5, 8, 10, 12, 13):
        N = 1 << M
        for code in ("const0", "freezer"):
            fro = oracle.const0(M, N // 2)[0] if code == "const0" else oracle.freezer(M, 0.5, 0.5)[0]
            K = int((fro == 0).sum())
            prog = oracle.compile(fro, M)
            cw = oracle.encode(pm1_message(M, K, 19), fro, M, 0)
            for fam in ("awgn-1", "uniform", "small"):
                llr = family_llr(fam, 5 * M + 1, cw).astype(np.float32)
                got, exp = oracle.decode_f32(llr, prog), ref.decode_f32(llr, prog, K)
                assert got.tobytes() == exp.tobytes(), (M, code, fam)
            frac = (rand_bytes(M, (N, 19)).astype(np.float32) - 127.5) / 16.0 * cw
            assert oracle.decode_f32(frac, prog).tobytes() == ref.decode_f32(frac, prog, K).tobytes(), (M, code, "fractional")
