"""polar_hip_decode_static (include/polar_hip.h): the entry point that allocates nothing, called
through the C ABI and compared bit for bit with the oracle (and with polar_hip_decode) on the same
inputs -- two calls in a row, the second with a smaller input (stale scratch of the first call must
not leak), codes above and below the LDS levels, and the documented rejections."""
import ctypes

import numpy as np
import pytest
import torch

from inputs import family_llr, pm1_message

pytestmark = pytest.mark.gpu


def _pair(session, oracle, M, frames, fam, seed):
    import polar_amd

    lib = polar_amd.lib
    N = 1 << M
    fro, _ = oracle.const0(M, N // 2)
    prog = oracle.compile(fro, M)
    pid, rows = session.load_program(prog)
    cw = oracle.encode(pm1_message(seed, N // 2, frames), fro, M, 0)
    host = family_llr(fam, seed + 1, cw)
    llr = torch.from_numpy(host).cuda()
    ref = torch.full((rows, frames), 55, dtype=torch.int8, device="cuda")
    got = torch.full((rows, frames), 66, dtype=torch.int8, device="cuda")
    torch.cuda.synchronize()
    assert lib.polar_hip_decode(session._h, pid, ctypes.c_void_p(ref.data_ptr()), ctypes.c_void_p(llr.data_ptr()), frames, 1) == 0, lib.polar_hip_last_error()
    free_before = torch.cuda.mem_get_info()[0]
    rc = lib.polar_hip_decode_static(session._h, pid, ctypes.c_void_p(got.data_ptr()), ctypes.c_void_p(llr.data_ptr()), frames, ctypes.c_void_p(0))
    assert rc == 0, lib.polar_hip_last_error()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before  # nothing was allocated on the device
    assert (got.cpu().numpy() == oracle.decode(host, prog)).all(), (M, frames, fam)  # the oracle, not only HIP against HIP
    return ref, got


def test_static_entry_point_matches_decode(session, oracle):
    # first call: N = 2^14 (levels 9..13 in the static array), many groups per wavefront; 700 groups
    # are whole teams of four (the team kernel), 701 are not (single wavefronts)
    ref, got = _pair(session, oracle, 14, 32 * 700, "uniform", 3)
    assert bool((ref == got).all())
    ref, got = _pair(session, oracle, 14, 32 * 701, "uniform", 4)
    assert bool((ref == got).all())
    # second call, smaller input and smaller code, same array
    ref, got = _pair(session, oracle, 12, 32 * 9, "small", 5)
    assert bool((ref == got).all())
    # a code that lives in LDS alone, and the full-size code (few resident wavefronts)
    ref, got = _pair(session, oracle, 8, 64, "awgn-1", 7)
    assert bool((ref == got).all())
    ref, got = _pair(session, oracle, 16, 32 * 40, "uniform", 9)
    assert bool((ref == got).all())


def test_static_entry_point_rejections(session, oracle):
    import polar_amd

    lib = polar_amd.lib
    M = 10
    N = 1 << M
    fro, _ = oracle.const0(M, N // 2)
    prog = oracle.compile(fro, M)
    pid, rows = session.load_program(prog)
    llr = torch.zeros((N, 40), dtype=torch.int8, device="cuda")
    out = torch.zeros((rows, 40), dtype=torch.int8, device="cuda")
    rc = lib.polar_hip_decode_static(session._h, pid, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(llr.data_ptr()), 40, ctypes.c_void_p(0))
    assert rc == -1 and b"multiple of 32" in lib.polar_hip_last_error()
    rc = lib.polar_hip_decode_static(session._h, pid + 1000, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(llr.data_ptr()), 32, ctypes.c_void_p(0))
    assert rc == -1
    # one wavefront's arena is 40 * 2^level bytes (32 frames, LLR levels + two planes): N = 2^20 needs
    # 40 MiB and is accepted, N = 2^21 needs 80 MiB, beyond the 64 MiB array. Rate-0 codes: they decode
    # nothing, but the arena is sized by the level.
    tiny = torch.zeros(32, dtype=torch.int8, device="cuda")
    for level, ok in ((20, True), (21, False)):
        big = np.array([level, 3, 255], np.uint8)
        pid2 = lib.polar_hip_load_program(session._h, big.ctypes.data, big.size)
        assert pid2 >= 0, lib.polar_hip_last_error()
        wide = torch.zeros((1 << level, 32), dtype=torch.int8, device="cuda")
        rc = lib.polar_hip_decode_static(session._h, pid2, ctypes.c_void_p(tiny.data_ptr()), ctypes.c_void_p(wide.data_ptr()), 32, ctypes.c_void_p(0))
        torch.cuda.synchronize()
        if ok:
            assert rc == 0, lib.polar_hip_last_error()
        else:
            assert rc == -1 and b"static array" in lib.polar_hip_last_error()
        del wide


def test_static_entry_point_in_a_hip_graph(session, oracle):
    """One of the callers the entry point is for: the launch is captured into a HIP graph (a capture
    tolerates neither allocation nor synchronisation) and replayed on new LLRs in the same buffers."""
    import polar_amd

    lib = polar_amd.lib
    M, frames = 12, 32 * 24
    N = 1 << M
    fro, _ = oracle.const0(M, N // 2)
    prog = oracle.compile(fro, M)
    pid, rows = session.load_program(prog)
    llr = torch.zeros((N, frames), dtype=torch.int8, device="cuda")
    out = torch.zeros((rows, frames), dtype=torch.int8, device="cuda")
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        rc = lib.polar_hip_decode_static(session._h, pid, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(llr.data_ptr()), frames, ctypes.c_void_p(side.cuda_stream))
        assert rc == 0, lib.polar_hip_last_error()
    for seed, fam in ((11, "uniform"), (12, "awgn-1")):
        cw = oracle.encode(pm1_message(seed, N // 2, frames), fro, M, 0)
        host = family_llr(fam, seed + 1, cw)
        llr.copy_(torch.from_numpy(host))
        out.fill_(77)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert (out.cpu().numpy() == oracle.decode(host, prog)).all(), fam
