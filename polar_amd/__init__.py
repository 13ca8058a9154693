"""polar_amd -- MI355X (gfx950) native polar-code SC decoding behind the xdsopl/polar API.

The compute path is the hand-written HIP library polar_amd/libpolar_hip.so (sources in
polar_amd/csrc/, C ABI in include/polar_hip.h). This package is only the host-side mirror of
the reference's functor interface (PolarEncoder / PolarSysEnc / PolarDecoder / PolarCompiler /
PolarFreezer / PolarCodeConst0) on top of that ABI; it contains no arithmetic of its own and
no CPU fallback: importing it without the built library raises.
"""
from .api import (  # noqa: F401
    DEVICE,
    HOST,
    HostBuffer,
    PolarCodeConst0,
    PolarCompiler,
    PolarDecoder,
    PolarEncoder,
    PolarError,
    PolarFreezer,
    PolarSysEnc,
    Session,
    pitched_empty,
    testbench_probability,
)
from ._lib import LIB_PATH, lib  # noqa: F401
