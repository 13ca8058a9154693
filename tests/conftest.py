import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _make(target_dir, *args):
    subprocess.run(["make", "-C", os.path.join(ROOT, target_dir), *args], check=True, capture_output=True)


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib

    if not os.path.exists(oracle_lib.ORACLE_SO):
        _make("oracle", "liboracle.so")
    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def ref():
    """The genuine reference build; present in the build container and shipped prebuilt."""
    import oracle_lib

    if not os.path.exists(oracle_lib.REF_SO):
        if os.path.isdir("/root/reference"):
            _make("oracle", "ref")
        else:
            pytest.skip("oracle/_ref/libpolar_ref.so not built and /root/reference absent")
    return oracle_lib.Ref()


@pytest.fixture(scope="session")
def golden():
    g = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(g, "polar_golden.json")) as f:
        meta = json.load(f)
    arrays = np.load(os.path.join(g, "polar_golden.npz"), allow_pickle=False)
    return meta, arrays


@pytest.fixture(scope="session")
def emu():
    """Host emulation build of the wavefront decoder core (tests/emu)."""
    import ctypes

    so = os.path.join(ROOT, "tests", "emu", "libemu.so")
    src = os.path.join(ROOT, "tests", "emu", "emu_decode.cc")
    deps = [src] + [os.path.join(ROOT, "polar_amd", "csrc", f) for f in ("sc_core.hpp", "program.hpp", "construct.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-shared", "-fPIC", "-o", so, src], check=True)
    E = ctypes.CDLL(so)
    E.emu_decode.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]
    E.emu_op.argtypes = [ctypes.c_int] + [ctypes.c_uint] * 4
    E.emu_op.restype = ctypes.c_uint
    E.emu_op_count.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    return E


@pytest.fixture(scope="session")
def session():
    """One device context for the GPU tests."""
    import polar_amd

    s = polar_amd.Session(0)
    yield s
    s.close()
