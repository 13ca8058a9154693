"""Host-side mirror of the reference's functor interface over the C ABI (include/polar_hip.h).

Same names and argument meaning as the reference classes; buffers are numpy arrays (host) or
torch CUDA tensors / raw device pointers (device) in the reference's element-major /
frame-minor layout, shape [elements][frames] (testbench.cc:134-139). Where the reference
asserts, these raise PolarError.
"""
import ctypes

import numpy as np

from ._lib import lib

HOST, DEVICE = 0, 1


class PolarError(RuntimeError):
    def __init__(self, code, where):
        msg = lib.polar_hip_last_error()
        super().__init__("%s failed (%d): %s" % (where, code, msg.decode() if msg else ""))
        self.code = code


def _check(rc, where):
    if rc < 0:
        raise PolarError(rc, where)
    return rc


def _ptr(a):
    """(pointer, where) of a numpy array, a torch tensor or None."""
    if a is None:
        return None, None
    if isinstance(a, np.ndarray):
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("buffers must be C-contiguous [elements][frames]")
        return ctypes.c_void_p(a.ctypes.data), HOST
    if hasattr(a, "data_ptr"):  # torch tensor
        if not a.is_contiguous():
            raise ValueError("buffers must be contiguous [elements][frames]")
        return ctypes.c_void_p(a.data_ptr()), DEVICE if a.is_cuda else HOST
    raise TypeError("expected a numpy array or a torch tensor")


def _pitched(a):
    """Row pitch in bytes of a 2-D int8 device tensor whose rows are contiguous but further apart than their
    length (a column range of a wider tensor, pitched_empty), or None."""
    if a is None or isinstance(a, np.ndarray) or not hasattr(a, "stride") or not a.is_cuda or a.dim() != 2 or a.is_contiguous():
        return None
    if a.element_size() != 1 or a.stride(1) != 1 or a.stride(0) < a.shape[1]:
        return None
    return int(a.stride(0))


def pitched_empty(rows, frames, device="cuda"):
    """int8 [rows][frames] device tensor with the row pitch the library suggests for decoder input
    (polar_hip_row_pitch): a view of a [rows][pitch] allocation."""
    import torch

    pitch = int(lib.polar_hip_row_pitch(int(frames)))
    return torch.empty((rows, pitch), dtype=torch.int8, device=device)[:, :frames]


def _where(*ws):
    ws = [w for w in ws if w is not None]
    if any(w != ws[0] for w in ws):
        raise ValueError("all buffers of one call must live on the same side (host or device)")
    return ws[0] if ws else HOST


class HostBuffer:
    """Page-locked host memory as a numpy array (polar_hip_host_alloc): frame buffers that
    PolarDecoder copies over PCIe while it decodes. `array` is valid until close()."""

    def __init__(self, shape, dtype=np.int8):
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._p = lib.polar_hip_host_alloc(max(nbytes, 1))
        if not self._p:
            raise PolarError(-4, "polar_hip_host_alloc")
        self.array = np.ctypeslib.as_array((ctypes.c_uint8 * max(nbytes, 1)).from_address(self._p))[:nbytes].view(dtype).reshape(shape)

    def close(self):
        if self._p:
            self.array = None
            lib.polar_hip_host_free(ctypes.c_void_p(self._p))
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def testbench_probability():
    """Erasure probability the reference bench feeds PolarCodeConst0 for rate 1/2 (testbench.cc:74-87)."""
    import math

    design = 10.0 * math.log10(-math.log(0.5))
    return math.exp(-(10.0 ** ((design + 1.59175) / 10.0)))


class PolarCodeConst0:
    """polar_freezer.hh:34-62. __call__(frozen, level, K, erasure_probability=exp(-1))."""

    def __call__(self, frozen_bits, level, K, erasure_probability=0.36787944117144233):
        assert frozen_bits.dtype == np.uint8 and frozen_bits.size >= (1 << level)
        _check(lib.polar_code_const0(frozen_bits.ctypes.data, level, K, erasure_probability), "PolarCodeConst0")


    def ambiguous(self, frozen_bits, level, K, erasure_probability=0.36787944117144233):
        """The construction, returning whether the K-boundary fell on an exact tie (the reference's
        std::nth_element leaves that case implementation-defined, polar_freezer.hh:56)."""
        amb = ctypes.c_int(0)
        _check(lib.polar_code_const0_ex(frozen_bits.ctypes.data, level, K, erasure_probability, ctypes.byref(amb)), "PolarCodeConst0")
        return bool(amb.value)

    def batch(self, level, K, erasure_probabilities):
        """One frozen set per erasure probability (design SNR), built on the host cores in parallel.
        Returns (frozen [count][N] uint8, ambiguous [count] bool)."""
        ps = (ctypes.c_longdouble * len(erasure_probabilities))(*erasure_probabilities)
        out = np.zeros((len(erasure_probabilities), 1 << level), np.uint8)
        amb = np.zeros(len(erasure_probabilities), np.int32)
        _check(lib.polar_code_const0_batch(out.ctypes.data, amb.ctypes.data, level, K, ctypes.cast(ps, ctypes.c_void_p), len(erasure_probabilities)), "PolarCodeConst0.batch")
        return out, amb.astype(bool)


class PolarFreezer:
    """polar_freezer.hh:11-32. __call__(frozen, level, erasure_probability=0.5, threshold=0.5) -> K."""

    def __call__(self, frozen_bits, level, erasure_probability=0.5, freezing_threshold=0.5):
        assert frozen_bits.dtype == np.uint8 and frozen_bits.size >= (1 << level)
        return _check(lib.polar_freezer(frozen_bits.ctypes.data, level, erasure_probability, freezing_threshold), "PolarFreezer")


class PolarCompiler:
    """polar_compiler.hh:9-59. __call__(program, frozen, level) -> length in bytes."""

    def __call__(self, program, frozen, level):
        assert program.dtype == np.uint8 and frozen.dtype == np.uint8
        if program.size < 2 * (1 << level) + 1:
            # the reference trusts the caller (testbench.cc:95 gives N bytes); refuse what could overflow
            probe = np.zeros(2 * (1 << level) + 1, np.uint8)
            n = _check(lib.polar_compile(probe.ctypes.data, frozen.ctypes.data, level), "PolarCompiler")
            if n > program.size:
                raise ValueError("program buffer too small: need %d bytes" % n)
            program[:n] = probe[:n]
            return n
        return _check(lib.polar_compile(program.ctypes.data, frozen.ctypes.data, level), "PolarCompiler")


class Session:
    """Owns one device context (decoder scratch, resident programs), like the reference's
    heap-allocated functor objects (testbench.cc:83,100)."""

    def __init__(self, device=0):
        h = ctypes.c_void_p()
        _check(lib.polar_hip_create(ctypes.byref(h), device), "polar_hip_create")
        self._h = h
        self._programs = {}
        self._stream = None  # None: the handle's private stream

    def close(self):
        if getattr(self, "_h", None):
            lib.polar_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _order(self, *arrays):
        """Device tensors are produced on torch's current stream: make it the handle's stream so
        the library's work queues behind the producers with no host synchronisation
        (include/polar_hip.h, stream ordering). Calls without a device tensor go back to the handle's
        private stream, so host-buffer work never lands on a stream the caller has meanwhile
        destroyed or on torch's blocking null stream. The stream in use must outlive the call;
        inputs produced on other streams than torch's current one are the caller's to synchronise."""
        for a in arrays:
            if a is not None and hasattr(a, "is_cuda") and a.is_cuda:
                import torch

                st = torch.cuda.current_stream(a.device).cuda_stream
                if st != self._stream:
                    _check(lib.polar_hip_set_stream(self._h, ctypes.c_void_p(st), 1), "polar_hip_set_stream")
                    self._stream = st
                return
        if self._stream is not None:
            _check(lib.polar_hip_set_stream(self._h, None, 0), "polar_hip_set_stream")
            self._stream = None

    def configure(self, frames_per_wave=0, lds_level=0, max_waves=0, team=1):
        _check(lib.polar_hip_configure_team(self._h, frames_per_wave, lds_level, team, max_waves), "polar_hip_configure")

    def load_program(self, program):
        program = np.ascontiguousarray(program, np.uint8)
        key = program.tobytes()
        if key not in self._programs:
            pid = _check(lib.polar_hip_load_program(self._h, program.ctypes.data, program.size), "polar_hip_load_program")
            rows = _check(lib.polar_program_rows(program.ctypes.data, program.size), "polar_program_rows")
            self._programs[key] = (pid, rows)
        return self._programs[key]

    def code(self, level, K, erasure_probability):
        """Cached (level, K, p) -> (frozen, program, ambiguous): construction, compilation and upload
        happen once per key inside the handle (include/polar_hip.h, polar_hip_code)."""
        fz, pr, n, amb = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int(0), ctypes.c_int(0)
        pid = _check(lib.polar_hip_code(self._h, level, K, erasure_probability, ctypes.byref(fz), ctypes.byref(pr), ctypes.byref(n), ctypes.byref(amb)), "polar_hip_code")
        frozen = np.ctypeslib.as_array(ctypes.cast(fz, ctypes.POINTER(ctypes.c_uint8)), shape=(1 << level,)).copy()
        program = np.ctypeslib.as_array(ctypes.cast(pr, ctypes.POINTER(ctypes.c_uint8)), shape=(n.value,)).copy()
        rows = _check(lib.polar_program_rows(program.ctypes.data, program.size), "polar_program_rows")
        self._programs[program.tobytes()] = (pid, rows)
        return frozen, program, bool(amb.value)

    def count(self, counters, noisy, orig, decoded, message):
        """testbench.cc:185-192: counters[4] (int64) += awgn_errors, quantization_erasures,
        uncorrected_errors, ambiguity_erasures. Any of the two pairs may be None."""
        self._order(counters, noisy, decoded)
        ptrs = [_ptr(x) for x in (counters, noisy, orig, decoded, message)]
        where = _where(*[w for _p, w in ptrs])
        n_code = int(noisy.numel() if hasattr(noisy, "numel") else noisy.size) if noisy is not None else 0
        n_mesg = int(decoded.numel() if hasattr(decoded, "numel") else decoded.size) if decoded is not None else 0
        _check(lib.polar_hip_count(self._h, ptrs[0][0], ptrs[1][0], ptrs[2][0], n_code, ptrs[3][0], ptrs[4][0], n_mesg, where), "polar_hip_count")

    def random_message(self, message, seed, first_frame=0):
        """testbench.cc:126-127 with the counter-based generator of include/polar_hip.h:
        message[K][frames] = 1 - 2 * bit for global frames first_frame, first_frame + 1, ..."""
        self._order(message)
        p, where = _ptr(message)
        _check(lib.polar_hip_random_message(self._h, p, int(message.shape[0]), _frames(message), int(seed), int(first_frame), where), "polar_hip_random_message")

    def channel(self, llr, codeword, snr_db, seed, first_frame=0):
        """testbench.cc:112-114, 151-165: BPSK + AWGN at Es/N0 = snr_db, LLR quantised to int8."""
        self._order(llr)
        (pl, wl), (pc, wc) = _ptr(llr), _ptr(codeword)
        if tuple(llr.shape) != tuple(codeword.shape):
            raise PolarError(-1, "channel: llr and codeword must have the same shape")
        if _is_f32(llr) or _is_f32(codeword):  # the float bench: unquantised LLRs (testbench.cc:49-53)
            if not (_is_f32(llr) and _is_f32(codeword)):
                raise ValueError("the float channel needs float32 llr and codeword buffers")
            _check(lib.polar_hip_channel_f32(self._h, pl, pc, int(codeword.shape[0]), _frames(codeword), float(snr_db), int(seed), int(first_frame), _where(wl, wc)), "polar_hip_channel_f32")
            return
        _check(lib.polar_hip_channel(self._h, pl, pc, int(codeword.shape[0]), _frames(codeword), float(snr_db), int(seed), int(first_frame), _where(wl, wc)), "polar_hip_channel")

    def simulate(self, counters, program, frozen, level, systematic, snr_db, seed, first_frame, frames, code_type="int8"):
        """One SNR point of testbench.cc:125-193 on the device; counters: int64[6] (host numpy or
        device tensor), added to. code_type "float": the bench's `typedef float code_type`
        (testbench.cc:49-53), same frames, unquantised LLRs, float decoder."""
        pid, _rows = self.load_program(program)
        fz = np.ascontiguousarray(frozen, dtype=np.uint8)
        self._order(counters)
        pc, where = _ptr(counters)
        if code_type not in ("int8", "float"):
            raise ValueError("code_type must be 'int8' or 'float'")
        fn = lib.polar_hip_simulate if code_type == "int8" else lib.polar_hip_simulate_f32
        _check(fn(self._h, pid, fz.ctypes.data, int(level), int(bool(systematic)), float(snr_db), int(seed), int(first_frame), int(frames), pc, where), "polar_hip_simulate")

    def unpack_symbols(self, symbols, sign_plane, erasure_plane=None):
        """symbols[rows][frames] (+1 / -1 / 0) from the bit planes of extract_systematic_packed."""
        self._order(symbols)
        (yp, yw), (sp, sw), (ep, ew) = _ptr(symbols), _ptr(sign_plane), _ptr(erasure_plane)
        _check(lib.polar_hip_unpack_symbols(self._h, yp, sp, ep, int(symbols.shape[0]), _frames(symbols), _where(yw, sw, ew)), "polar_hip_unpack_symbols")

    def synchronize(self):
        _check(lib.polar_hip_synchronize(self._h), "polar_hip_synchronize")

    def last_decode_shape(self):
        """(frames per wavefront, LDS level, team) of the last decode launch."""
        f, r, t = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _check(lib.polar_hip_last_decode_shape(self._h, ctypes.byref(f), ctypes.byref(r), ctypes.byref(t)), "polar_hip_last_decode_shape")
        return f.value, r.value, t.value

    def last_decode_launches(self):
        """Kernel launches the last decode went out as (relay launches of whole sweeps; 1 = a single launch)."""
        n = lib.polar_hip_last_decode_launches(self._h)
        _check(min(n, 0), "polar_hip_last_decode_launches")
        return n

    def last_decode_ms(self):
        ms = lib.polar_hip_last_decode_ms(self._h)
        if ms < 0:
            raise PolarError(int(ms), "polar_hip_last_decode_ms")
        return ms


_default_session = None


def _session(session):
    global _default_session
    if session is not None:
        return session
    if _default_session is None:
        _default_session = Session(0)
    return _default_session


def _is_f32(a):
    return str(a.dtype) in ("float32", "torch.float32")


def _frames(a):
    return int(a.shape[1]) if a.ndim == 2 else 1


class PolarDecoder:
    """polar_decoder.hh:9-465: __call__(message, codeword, program)."""

    def __init__(self, session=None):
        self.session = _session(session)

    def __call__(self, message, codeword, program):
        pid, rows = self.session.load_program(program)
        frames = _frames(codeword)
        n = 1 << int(np.asarray(program).ravel()[0])
        if codeword.shape[0] != n or (rows and (message.shape[0] < rows or _frames(message) != frames)):
            raise ValueError("buffer shapes do not match the program: codeword [%d][F], message [>=%d][F]" % (n, rows))
        self.session._order(codeword)
        cpitch, mpitch = _pitched(codeword), _pitched(message)
        if cpitch or mpitch:  # rows further apart than their length: device tensors, used in place
            if not (getattr(codeword, "is_cuda", False) and getattr(message, "is_cuda", False)):
                raise ValueError("pitched buffers must both be device tensors")
            if (not cpitch and not codeword.is_contiguous()) or (not mpitch and not message.is_contiguous()):
                raise ValueError("buffers must be [elements][frames] with contiguous rows")
            _check(lib.polar_hip_decode_pitched(self.session._h, pid, ctypes.c_void_p(message.data_ptr()), mpitch or 0,
                                                ctypes.c_void_p(codeword.data_ptr()), cpitch or 0, frames), "PolarDecoder(pitched)")
            return
        mp, mw = _ptr(message)
        cp, cw = _ptr(codeword)
        if _is_f32(codeword) or _is_f32(message):  # TYPE = float lanes (testbench.cc:49-53): the reference's float helper
            if not (_is_f32(codeword) and _is_f32(message)):
                raise ValueError("float decoding needs float32 codeword and message buffers")
            _check(lib.polar_hip_decode_f32(self.session._h, pid, mp, cp, frames, _where(mw, cw)), "PolarDecoder(float)")
            return
        _check(lib.polar_hip_decode(self.session._h, pid, mp, cp, frames, _where(mw, cw)), "PolarDecoder")

    def decode_set(self, jobs):
        """polar_hip_decode_set: jobs = [(message, codeword, program), ...], device tensors of any mix of codes
        and batch sizes, decoded side by side on the handle's internal lanes; returns when all are done."""

        class Job(ctypes.Structure):
            _fields_ = [("program_id", ctypes.c_int), ("message", ctypes.c_void_p), ("codeword", ctypes.c_void_p), ("frames", ctypes.c_size_t),
                        ("message_pitch", ctypes.c_size_t), ("codeword_pitch", ctypes.c_size_t)]

        arr = (Job * len(jobs))()
        for i, (message, codeword, program) in enumerate(jobs):
            pid, rows = self.session.load_program(program)
            n = 1 << int(np.asarray(program).ravel()[0])
            frames = _frames(codeword)
            if codeword.shape[0] != n or (rows and (message.shape[0] < rows or _frames(message) != frames)):
                raise ValueError("job %d: buffer shapes do not match the program" % i)
            cpitch, mpitch = _pitched(codeword), _pitched(message)  # rows further apart than their length: used in place
            (mp, mw) = (ctypes.c_void_p(message.data_ptr()), DEVICE) if mpitch else _ptr(message)
            (cp, cw) = (ctypes.c_void_p(codeword.data_ptr()), DEVICE) if cpitch else _ptr(codeword)
            if _where(mw, cw) != DEVICE:
                raise ValueError("job %d: the batches of a set are device tensors" % i)
            arr[i] = Job(pid, mp.value, cp.value, frames, mpitch or 0, cpitch or 0)
        if jobs:
            self.session._order(jobs[0][1])
        _check(lib.polar_hip_decode_set(self.session._h, ctypes.cast(arr, ctypes.c_void_p), len(jobs)), "PolarDecoder.decode_set")

    def launch(self, message, codeword, program, stream):
        """Enqueue on a HIP stream without synchronising (device buffers only)."""
        pid, _ = self.session.load_program(program)
        mp, _w = _ptr(message)
        cp, _w = _ptr(codeword)
        _check(lib.polar_hip_decode_async(self.session._h, pid, mp, cp, _frames(codeword), ctypes.c_void_p(stream)), "PolarDecoder.launch")


class _Encoder:
    systematic = 0

    def __init__(self, M, session=None):
        self.M = M
        self.session = _session(session)

    def __call__(self, codeword, message, frozen):
        frozen = np.ascontiguousarray(frozen, np.uint8)
        if codeword.shape[0] != (1 << self.M) or frozen.size != (1 << self.M):
            raise ValueError("codeword / frozen must have N = %d rows" % (1 << self.M))
        self.session._order(codeword)
        cp, cw = _ptr(codeword)
        mp, mw = _ptr(message)
        if _is_f32(codeword) or (message is not None and _is_f32(message)):  # TYPE = float lanes (testbench.cc:49-53)
            if not (_is_f32(codeword) and (message is None or _is_f32(message))):
                raise ValueError("float encoding needs float32 codeword and message buffers")
            _check(lib.polar_hip_encode_f32(self.session._h, self.systematic, cp, mp, frozen.ctypes.data, self.M, _frames(codeword), _where(cw, mw)), type(self).__name__ + "(float)")
            return
        _check(lib.polar_hip_encode(self.session._h, self.systematic, cp, mp, frozen.ctypes.data, self.M, _frames(codeword), _where(cw, mw)), type(self).__name__)


class PolarEncoder(_Encoder):
    """polar_encoder.hh:9-28: __call__(codeword, message, frozen)."""

    def extract_systematic(self, decoded, frozen):
        """testbench.cc:177-183: re-encode `decoded` and gather the information positions, in place."""
        frozen = np.ascontiguousarray(frozen, np.uint8)
        self.session._order(decoded)
        dp, dw = _ptr(decoded)
        _check(lib.polar_hip_systematic_extract(self.session._h, dp, frozen.ctypes.data, self.M, _frames(decoded), dw), "extract_systematic")


    def extract_systematic_packed(self, sign_plane, erasure_plane, decoded, frozen):
        """testbench.cc:177-183 with the bit-packed wire format of include/polar_hip.h: planes are
        uint8 [K][4 * ceil(frames / 32)]; erasure_plane may be None; `decoded` is left alone."""
        frozen = np.ascontiguousarray(frozen, np.uint8)
        self.session._order(decoded)
        (sp, sw), (ep, ew), (dp, dw) = _ptr(sign_plane), _ptr(erasure_plane), _ptr(decoded)
        _check(lib.polar_hip_systematic_extract_packed(self.session._h, sp, ep, dp, frozen.ctypes.data, self.M, _frames(decoded), _where(sw, ew, dw)), "extract_systematic_packed")


class PolarSysEnc(_Encoder):
    """polar_encoder.hh:30-59: __call__(codeword, message, frozen)."""

    systematic = 1
