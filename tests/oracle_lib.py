"""ctypes bindings for the CPU checkers under oracle/ (TEST INFRASTRUCTURE ONLY).

`oracle`  : our plain-C restatement (oracle/liboracle.so)
`ref`     : the genuine reference headers behind oracle/ref_shim.cc
            (oracle/_ref/libpolar_ref.so; built in the build container, travels prebuilt)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
All arrays are numpy, layout [elem][frame] (element-major / frame-minor, testbench.cc:134-139).
"""
import ctypes
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(_ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(_ROOT, "oracle", "_ref", "libpolar_ref.so")

_vp = ctypes.c_void_p
_i = ctypes.c_int
_l = ctypes.c_long
_sz = ctypes.c_size_t
_ld = ctypes.c_longdouble
_i8 = ctypes.c_int8


def _ptr(a):
    return a.ctypes.data_as(_vp)


def _c(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


class Oracle:
    def __init__(self, path=ORACLE_SO):
        L = ctypes.CDLL(path)
        self.L = L
        for name in ("qabs", "signum", "decide"):
            f = getattr(L, "oracle_" + name)
            f.argtypes, f.restype = [_i8], _i8
        for name in ("qmin", "qadd", "qmul", "prod"):
            f = getattr(L, "oracle_" + name)
            f.argtypes, f.restype = [_i8, _i8], _i8
        L.oracle_madd.argtypes, L.oracle_madd.restype = [_i8, _i8, _i8], _i8
        L.oracle_flip.argtypes, L.oracle_flip.restype = [_i8, _i8, _i8, _i8], _i8
        L.oracle_quant.argtypes, L.oracle_quant.restype = [ctypes.c_double], _i8
        L.oracle_const0.argtypes, L.oracle_const0.restype = [_vp, _i, _i, _ld], _i
        L.oracle_freezer.argtypes, L.oracle_freezer.restype = [_vp, _i, _ld, _ld], _i
        L.oracle_testbench_probability.argtypes, L.oracle_testbench_probability.restype = [], _ld
        L.oracle_compile.argtypes, L.oracle_compile.restype = [_vp, _vp, _i], _i
        L.oracle_encode.argtypes, L.oracle_encode.restype = [_vp, _vp, _vp, _i, _sz, _i], None
        L.oracle_decode.argtypes, L.oracle_decode.restype = [_vp, _vp, _vp, _sz], _l
        L.oracle_systematic_extract.argtypes, L.oracle_systematic_extract.restype = [_vp, _vp, _i, _sz], None
        L.oracle_count.argtypes, L.oracle_count.restype = [_vp, _vp, _vp, _sz, _vp, _vp, _sz], None
        L.oracle_decode_f32.argtypes, L.oracle_decode_f32.restype = [_vp, _vp, _vp, _sz], _l
        L.oracle_decode_time.argtypes, L.oracle_decode_time.restype = [_vp, _vp, _sz, _i], ctypes.c_double

    def const0(self, level, K, p=None):
        frozen = np.zeros(1 << level, np.uint8)
        if p is None:
            p = self.L.oracle_testbench_probability()
        amb = self.L.oracle_const0(_ptr(frozen), level, K, p)
        return frozen, amb

    def freezer(self, level, p=0.5, th=0.5):
        frozen = np.zeros(1 << level, np.uint8)
        K = self.L.oracle_freezer(_ptr(frozen), level, p, th)
        return frozen, K

    def compile(self, frozen, level):
        frozen = _c(frozen, np.uint8)
        prog = np.zeros(2 * (1 << level) + 2, np.uint8)  # worst case 2N+1 bytes
        n = self.L.oracle_compile(_ptr(prog), _ptr(frozen), level)
        if n < 0:
            raise ValueError("frozen set not compilable (reference would assert)")
        return prog[:n].copy()

    def encode(self, message, frozen, level, systematic):
        frozen = _c(frozen, np.uint8)
        message = _c(message, np.int8)
        frames = message.shape[1] if message.ndim == 2 else 1
        cw = np.zeros(((1 << level), frames), np.int8)
        self.L.oracle_encode(_ptr(cw), _ptr(message), _ptr(frozen), level, frames, int(systematic))
        return cw

    def decode(self, codeword, program):
        program = _c(program, np.uint8)
        codeword = _c(codeword, np.int8)
        n = 1 << int(program[0])
        frames = codeword.shape[1]
        assert codeword.shape[0] == n
        msg = np.zeros((n, frames), np.int8)
        rows = self.L.oracle_decode(_ptr(msg), _ptr(codeword), _ptr(program), frames)
        if rows < 0:
            raise ValueError("malformed program")
        return msg[:rows].copy()

    def decode_f32(self, codeword, program):
        """PolarDecoder<SIMD<float,W>,M> (the reference's float path): float32 [N][frames] -> [rows][frames]."""
        program = _c(program, np.uint8)
        codeword = _c(codeword, np.float32)
        n = 1 << int(program[0])
        frames = codeword.shape[1]
        assert codeword.shape[0] == n
        msg = np.zeros((n, frames), np.float32)
        rows = self.L.oracle_decode_f32(_ptr(msg), _ptr(codeword), _ptr(program), frames)
        if rows < 0:
            raise ValueError("malformed program")
        return msg[:rows].copy()

    def systematic_extract(self, decoded, frozen, level):
        frozen = _c(frozen, np.uint8)
        d = np.array(decoded, dtype=np.int8, order="C", copy=True)
        self.L.oracle_systematic_extract(_ptr(d), _ptr(frozen), level, d.shape[1])
        return d

    def count(self, noisy, orig, decoded, message):
        c = np.zeros(4, np.int64)
        noisy, orig, decoded, message = (_c(x, np.int8) for x in (noisy, orig, decoded, message))
        self.L.oracle_count(_ptr(c), _ptr(noisy), _ptr(orig), noisy.size, _ptr(decoded), _ptr(message), decoded.size)
        return c

    def philox(self, counter, key):
        out = (ctypes.c_uint32 * 4)()
        self.L.oracle_philox4x32_10(out, (ctypes.c_uint32 * 4)(*counter), (ctypes.c_uint32 * 2)(*key))
        return list(out)

    def random_message(self, rows, frames, seed, first_frame=0):
        out = np.zeros((rows, frames), np.int8)
        self.L.oracle_random_message(_ptr(out), ctypes.c_size_t(rows), ctypes.c_size_t(frames), ctypes.c_uint64(seed), ctypes.c_uint64(first_frame))
        return out

    def channel(self, codeword, snr_db, seed, first_frame=0):
        codeword = _c(codeword, np.int8)
        out = np.zeros_like(codeword)
        self.L.oracle_channel(_ptr(out), _ptr(codeword), ctypes.c_size_t(codeword.shape[0]), ctypes.c_size_t(codeword.shape[1]), ctypes.c_double(snr_db), ctypes.c_uint64(seed), ctypes.c_uint64(first_frame))
        return out

    def channel_f32(self, codeword, snr_db, seed, first_frame=0):
        codeword = _c(codeword, np.int8)
        out = np.zeros(codeword.shape, np.float32)
        self.L.oracle_channel_f32(_ptr(out), _ptr(codeword), ctypes.c_size_t(codeword.shape[0]), ctypes.c_size_t(codeword.shape[1]), ctypes.c_double(snr_db), ctypes.c_uint64(seed), ctypes.c_uint64(first_frame))
        return out

    def quant(self, x):
        return np.array([self.L.oracle_quant(float(v)) for v in np.ravel(x)], np.int8).reshape(np.shape(x))

    def decode_time(self, codeword, program, reps):
        program = _c(program, np.uint8)
        codeword = _c(codeword, np.int8)
        return self.L.oracle_decode_time(_ptr(codeword), _ptr(program), codeword.shape[1], reps)


class Ref:
    """The genuine reference (compiled from /root/reference in the build container)."""

    def __init__(self, path=REF_SO):
        L = ctypes.CDLL(path)
        self.L = L
        L.ref_simd_width.restype = _i
        L.ref_const0.argtypes, L.ref_const0.restype = [_vp, _i, _i, _ld], None
        L.ref_const0_testbench.argtypes, L.ref_const0_testbench.restype = [_vp, _i, _i], None
        L.ref_freezer.argtypes, L.ref_freezer.restype = [_vp, _i, _ld, _ld], _i
        L.ref_compile.argtypes, L.ref_compile.restype = [_vp, _vp, _i], _i
        L.ref_encode.argtypes, L.ref_encode.restype = [_vp, _vp, _vp, _i, _i, _l, _i], None
        L.ref_decode.argtypes, L.ref_decode.restype = [_vp, _vp, _vp, _i, _l], None
        L.ref_decode_time.argtypes, L.ref_decode_time.restype = [_vp, _vp, _i], ctypes.c_double
        if hasattr(L, "ref_decode_f32"):
            L.ref_decode_f32.argtypes, L.ref_decode_f32.restype = [_vp, _vp, _vp, _i, _l], _i
        L.ref_op.argtypes, L.ref_op.restype = [_i, _i, _i, _i], _i
        L.ref_quant.argtypes, L.ref_quant.restype = [ctypes.c_double], _i
        self.width = L.ref_simd_width()

    def const0(self, level, K, p=None):
        frozen = np.zeros(1 << level, np.uint8)
        if p is None:
            self.L.ref_const0_testbench(_ptr(frozen), level, K)
        else:
            self.L.ref_const0(_ptr(frozen), level, K, p)
        return frozen

    def freezer(self, level, p=0.5, th=0.5):
        frozen = np.zeros(1 << level, np.uint8)
        K = self.L.ref_freezer(_ptr(frozen), level, p, th)
        return frozen, K

    def compile(self, frozen, level):
        frozen = _c(frozen, np.uint8)
        prog = np.zeros(2 * (1 << level) + 2, np.uint8)  # worst case 2N+1 bytes
        n = self.L.ref_compile(_ptr(prog), _ptr(frozen), level)
        return prog[:n].copy()

    def encode(self, message, frozen, level, systematic):
        frozen = _c(frozen, np.uint8)
        message = _c(message, np.int8)
        K, frames = message.shape
        cw = np.zeros(((1 << level), frames), np.int8)
        self.L.ref_encode(_ptr(cw), _ptr(message), _ptr(frozen), level, K, frames, int(systematic))
        return cw

    def decode(self, codeword, program, K):
        program = _c(program, np.uint8)
        codeword = _c(codeword, np.int8)
        frames = codeword.shape[1]
        msg = np.zeros((K, frames), np.int8)
        self.L.ref_decode(_ptr(msg), _ptr(codeword), _ptr(program), K, frames)
        return msg

    def decode_f32(self, codeword, program, K):
        program = _c(program, np.uint8)
        codeword = _c(codeword, np.float32)
        msg = np.zeros((K, codeword.shape[1]), np.float32)
        assert self.L.ref_decode_f32(_ptr(msg), _ptr(codeword), _ptr(program), K, codeword.shape[1]) == 0
        return msg

    def decode_time(self, codeword_bundle, program, reps):
        program = _c(program, np.uint8)
        cw = _c(codeword_bundle, np.int8)
        assert cw.shape[1] == self.width
        return self.L.ref_decode_time(_ptr(cw), _ptr(program), reps)

    OPS = {"prod": 0, "madd": 1, "qadd": 2, "qmul": 3, "qabs": 4, "signum": 5, "decide": 6, "qmin": 7, "flip": 8}

    def op(self, name, a, b=0, c=0):
        return self.L.ref_op(self.OPS[name], a, b, c)


def program_rows(program):
    """Number of message symbols a program emits (= K for a compiled frozen set)."""
    lvl = int(program[0])
    rows = 0
    for op in program[1:-1]:
        op = int(op)
        if op in (0, 7):
            lvl -= 1
        elif op in (2, 8):
            lvl += 1
        elif op == 4:
            rows += 1 << lvl
        elif op == 5:
            rows += 1
        elif op == 6:
            rows += (1 << lvl) - 1
        elif op == 9:
            lvl += 1
            rows += 1 << (lvl - 1)
    return rows


def fnv1a64(data):
    h = 0xCBF29CE484222325
    for x in bytes(data):
        h = ((h ^ x) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def awgn_llr(rng, codeword, snr_db):
    """BPSK/AWGN + LLR quantisation of testbench.cc:112-114,154-165 (numpy restatement)."""
    sigma = np.sqrt(1.0 / (2.0 * 10.0 ** (snr_db / 10.0)))
    symb = codeword.astype(np.float64) + rng.normal(0.0, sigma, codeword.shape)
    fact = 2.0 / (sigma * sigma)
    return np.clip(np.rint(fact * symb), -128, 127).astype(np.int8)
