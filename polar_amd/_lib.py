"""ctypes binding of libpolar_hip.so -- one declaration per symbol of include/polar_hip.h."""
import ctypes
import os

# POLAR_HIP_LIB: diagnostic builds of the same library (tools/), e.g. libpolar_hip_prof.so
LIB_PATH = os.environ.get("POLAR_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpolar_hip.so")

_vp = ctypes.c_void_p
_i = ctypes.c_int
_sz = ctypes.c_size_t
_ld = ctypes.c_longdouble

# every exported symbol with (argtypes, restype); tests check the library against this table
SYMBOLS = {
    "polar_hip_last_error": ([], ctypes.c_char_p),
    "polar_code_const0": ([_vp, _i, _i, _ld], _i),
    "polar_code_const0_ex": ([_vp, _i, _i, _ld, ctypes.POINTER(_i)], _i),
    "polar_code_const0_batch": ([_vp, _vp, _i, _i, _vp, _i], _i),
    "polar_freezer": ([_vp, _i, _ld, _ld], _i),
    "polar_compile": ([_vp, _vp, _i], _i),
    "polar_program_rows": ([_vp, _i], _i),
    "polar_hip_create": ([ctypes.POINTER(_vp), _i], _i),
    "polar_hip_destroy": ([_vp], None),
    "polar_hip_configure": ([_vp, _i, _i, _i], _i),
    "polar_hip_configure_team": ([_vp, _i, _i, _i, _i], _i),
    "polar_hip_last_decode_launches": ([_vp], _i),
    "polar_hip_decode_pitched": ([_vp, _i, _vp, _sz, _vp, _sz, _sz], _i),
    "polar_hip_row_pitch": ([_sz], _sz),
    "polar_hip_last_decode_shape": ([_vp, ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i)], _i),
    "polar_hip_set_stream": ([_vp, _vp, _i], _i),
    "polar_hip_load_program": ([_vp, _vp, _i], _i),
    "polar_hip_code": ([_vp, _i, _i, _ld, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_i), ctypes.POINTER(_i)], _i),
    "polar_hip_decode": ([_vp, _i, _vp, _vp, _sz, _i], _i),
    "polar_hip_where": ([_vp], _i),
    "polar_hip_host_alloc": ([_sz], _vp),
    "polar_hip_host_free": ([_vp], None),
    "polar_hip_decode_async": ([_vp, _i, _vp, _vp, _sz, _vp], _i),
    "polar_hip_decode_set": ([_vp, _vp, _i], _i),
    "polar_hip_decode_f32": ([_vp, _i, _vp, _vp, _sz, _i], _i),
    "polar_hip_encode_f32": ([_vp, _i, _vp, _vp, _vp, _i, _sz, _i], _i),
    "polar_hip_channel_f32": ([_vp, _vp, _vp, _sz, _sz, ctypes.c_double, ctypes.c_uint64, ctypes.c_uint64, _i], _i),
    "polar_hip_simulate_f32": ([_vp, _i, _vp, _i, _i, ctypes.c_double, ctypes.c_uint64, ctypes.c_uint64, _sz, _vp, _i], _i),
    "polar_hip_decode_static": ([_vp, _i, _vp, _vp, _sz, _vp], _i),
    "polar_hip_encode": ([_vp, _i, _vp, _vp, _vp, _i, _sz, _i], _i),
    "polar_hip_systematic_extract": ([_vp, _vp, _vp, _i, _sz, _i], _i),
    "polar_hip_systematic_extract_packed": ([_vp, _vp, _vp, _vp, _vp, _i, _sz, _i], _i),
    "polar_hip_unpack_symbols": ([_vp, _vp, _vp, _vp, _sz, _sz, _i], _i),
    "polar_hip_count": ([_vp, _vp, _vp, _vp, _sz, _vp, _vp, _sz, _i], _i),
    "polar_hip_random_message": ([_vp, _vp, _sz, _sz, ctypes.c_uint64, ctypes.c_uint64, _i], _i),
    "polar_hip_channel": ([_vp, _vp, _vp, _sz, _sz, ctypes.c_double, ctypes.c_uint64, ctypes.c_uint64, _i], _i),
    "polar_hip_simulate": ([_vp, _i, _vp, _i, _i, ctypes.c_double, ctypes.c_uint64, ctypes.c_uint64, _sz, _vp, _i], _i),
    "polar_hip_allreduce_counters": ([_vp, _vp, _sz, _vp], _i),
    "polar_hip_synchronize": ([_vp], _i),
    "polar_hip_last_decode_ms": ([_vp], ctypes.c_float),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "polar_amd: %s is missing -- build it with `make -C polar_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback." % LIB_PATH
        )
    L = ctypes.CDLL(LIB_PATH)
    for name, (argtypes, restype) in SYMBOLS.items():
        f = getattr(L, name)
        f.argtypes = argtypes
        f.restype = restype
    return L


lib = _load()
