"""GPU parity tests: the HIP path, called through the C ABI (include/polar_hip.h), against the
oracle on the same inputs and against the committed golden vectors from the reference.
Bit-exact: everything here is int8 / byte work."""
import numpy as np
import pytest
import torch

from inputs import corner_frames, family_llr, pm1_message, rand_bytes
from oracle_lib import fnv1a64

pytestmark = pytest.mark.gpu


def _code(oracle, M, code="const0", K=None):
    N = 1 << M
    if code == "const0":
        fro, _ = oracle.const0(M, N // 2 if K is None else K)
        return fro
    return oracle.freezer(M, 0.5, 0.5)[0]


def _decode(session, llr, prog, rows):
    import polar_amd

    out = np.zeros((max(rows, 1), llr.shape[1]), np.int8)
    polar_amd.PolarDecoder(session)(out, np.ascontiguousarray(llr), prog)
    return out[:rows]


def test_native_library_loaded(session):
    import polar_amd

    maps = open("/proc/self/maps").read()
    assert polar_amd.LIB_PATH in maps


def test_decoder_golden_kats(session, oracle, golden):
    meta, arr = golden
    for k in meta["decoder_kats"]:
        M = k["M"]
        N = 1 << M
        fro = _code(oracle, M, k["code"])
        K = int((fro == 0).sum())
        prog = oracle.compile(fro, M)
        cw = oracle.encode(pm1_message(k["msg_seed"], K, 8), fro, M, 0)
        llr = corner_frames(N) if k["family"] == "corner" else family_llr(k["family"], k["llr_seed"], cw)
        out = _decode(session, llr, prog, K)
        assert (out == arr[k["name"] + "_out"]).all(), k["name"]


@pytest.mark.parametrize("M", [12, 14, 16, 20])
def test_decoder_golden_digests(session, oracle, golden, M):
    meta, _ = golden
    N = 1 << M
    fro = _code(oracle, M)
    prog = oracle.compile(fro, M)
    for k in [k for k in meta["decoder_digests"] if k["M"] == M]:
        cw = oracle.encode(pm1_message(k["msg_seed"], N // 2, k["frames"]), fro, M, 1)
        llr = family_llr(k["family"], k["llr_seed"], cw)
        out = _decode(session, llr, prog, N // 2)
        assert "%016x" % fnv1a64(out) == k["out_fnv"], (M, k["family"])


@pytest.mark.parametrize("fpw,rl,team", [(4, 11, 1), (8, 10, 1), (16, 9, 1), (32, 8, 1), (64, 7, 1), (16, 10, 1), (16, 8, 1), (64, 6, 1), (128, 6, 1), (128, 5, 1), (256, 5, 1),
                                         (256, 4, 1), (32, 8, 4), (64, 7, 4)])
def test_decoder_vs_oracle_all_variants(session, oracle, fpw, rl, team):
    """Every built kernel variant, ragged frame counts (tail groups, unaligned rows, part teams)."""
    session.configure(fpw, rl, 0, team)
    try:
        for M, frames in ((1, 5), (2, 64), (3, 7), (5, 130), (8, 257), (10, 96), (11, 33), (13, 40)):
            N = 1 << M
            for code in ("const0", "freezer"):
                fro = _code(oracle, M, code)
                K = int((fro == 0).sum())
                if K in (0, N) and M > 2:
                    continue
                prog = oracle.compile(fro, M)
                cw = oracle.encode(pm1_message(M, max(K, 1), frames)[:K], fro, M, 0)
                for fam in ("awgn-1", "uniform", "small"):
                    llr = family_llr(fam, M + frames, cw)
                    assert (_decode(session, llr, prog, K) == oracle.decode(llr, prog)).all(), (M, frames, code, fam)
    finally:
        session.configure(0, 0, 0)


@pytest.mark.parametrize("M,fpw,rl,team", [(16, 128, 6, 1), (16, 128, 5, 1), (16, 64, 7, 1), (16, 256, 5, 1), (16, 64, 7, 4), (16, 32, 8, 4), (20, 32, 8, 4), (20, 64, 7, 4)])
def test_wide_variants_full_size_digest(session, oracle, golden, M, fpw, rl, team):
    """The variants the default picks for big batches, on the reference's N = 2^16 digests (and
    N = 2^20 for the team kernels, what config D runs): the 32 golden frames are embedded in a
    batch wide enough for whole 128/256-frame groups."""
    meta, _ = golden
    N = 1 << M
    fro = _code(oracle, M)
    prog = oracle.compile(fro, M)
    session.configure(fpw, rl, 0, team)
    try:
        for k in [k for k in meta["decoder_digests"] if k["M"] == M]:
            cw = oracle.encode(pm1_message(k["msg_seed"], N // 2, k["frames"]), fro, M, 1)
            llr = family_llr(k["family"], k["llr_seed"], cw)
            wide = np.ascontiguousarray(np.tile(llr, (1, 20))[:, :fpw * team + 32 + 5])  # ragged: one full group (team) + a tail
            out = _decode(session, wide, prog, N // 2)
            for a in range(0, wide.shape[1] - 31, 32):
                assert "%016x" % fnv1a64(np.ascontiguousarray(out[:, a:a + 32])) == k["out_fnv"], (fpw, rl, team, k["family"], a)
    finally:
        session.configure(0, 0, 0)


@pytest.mark.parametrize("M,frames", [(16, 1 << 19), (16, (1 << 18) + 128 * 300 + 52), (16, 40000 + 21), (20, 4096 + 96)])
def test_bench_grid_parity(session, oracle, golden, M, frames):
    """The exact path bench.py times for config B: default variant pick on a batch that fills every
    resident wavefront for several sweeps (N = 2^16, 2^19 frames: (128, 6), >= 2048 persistent
    workgroups, the XCD-aware group order, the multi-GiB scratch arena, the hot region), and the team
    kernels the default pick takes below that ((64, 7) and (32, 8) in teams of four) on sizes that
    are not a multiple of the sweep or of a team; N = 2^20 with 4192 frames takes a narrow
    single-wavefront shape (config D's own shape: test_config_d_shape_golden). The
    32 golden frames of the reference are tiled over the batch on the device: every 32-frame block
    of the output must equal block 0, and block 0 must hash to the reference's digest."""
    import polar_amd

    meta, _ = golden
    N = 1 << M
    fro = _code(oracle, M)
    prog = oracle.compile(fro, M)
    dec = polar_amd.PolarDecoder(session)
    out = torch.empty((N // 2, frames), dtype=torch.int8, device="cuda")
    for k in [k for k in meta["decoder_digests"] if k["M"] == M]:
        cw = oracle.encode(pm1_message(k["msg_seed"], N // 2, k["frames"]), fro, M, 1)
        llr = family_llr(k["family"], k["llr_seed"], cw)
        assert llr.shape[1] == 32
        d = torch.from_numpy(llr).cuda()
        wide = d.repeat(1, (frames + 31) // 32)[:, :frames].contiguous()
        del d
        out.fill_(77)
        dec(out, wide, prog)
        del wide
        whole = frames // 32 * 32
        blocks = out[:, :whole].view(N // 2, whole // 32, 32)
        assert bool((blocks == blocks[:, :1, :]).all()), (M, frames, k["family"])
        if frames > whole:
            assert bool((out[:, whole:] == out[:, :frames - whole]).all())
        assert "%016x" % fnv1a64(out[:, :32].contiguous().cpu().numpy()) == k["out_fnv"], (M, frames, k["family"])
    del out
    torch.cuda.empty_cache()


@pytest.mark.parametrize("mode", ["default-pick", "scratch-limited"])
def test_config_d_shape_golden(session, oracle, golden, mode):
    """Config D as bench.py runs it: N = 2^20 on the (32, 8) kernel in teams of four.
    default-pick: a batch large enough for the default to take that shape (>= 3/8 of the resident
    wavefronts get a group). scratch-limited: the same shape with so little free device memory that
    the arenas of all resident wavefronts do not fit (launch_decode then runs fewer workgroups for
    more sweeps) -- a ballast allocation takes the memory away. The reference's 32 golden frames are
    tiled over the batch: every 32-frame block must equal block 0 and hash to the reference's digest."""
    import polar_amd

    meta, _ = golden
    M = 20
    N = 1 << M
    fro = _code(oracle, M)
    prog = oracle.compile(fro, M)
    if mode == "scratch-limited":
        session = polar_amd.Session(0)  # a handle of its own: no scratch held from earlier tests
    dec = polar_amd.PolarDecoder(session)
    k = [k for k in meta["decoder_digests"] if k["M"] == M][0]
    cw = oracle.encode(pm1_message(k["msg_seed"], N // 2, k["frames"]), fro, M, 1)
    llr = family_llr(k["family"], k["llr_seed"], cw)
    frames = 24576 + 96 if mode == "default-pick" else 16384 + 128
    d = torch.from_numpy(llr).cuda()
    wide = d.repeat(1, (frames + 31) // 32)[:, :frames].contiguous()
    del d
    out = torch.full((N // 2, frames), 77, dtype=torch.int8, device="cuda")
    ballast = None
    try:
        if mode == "scratch-limited":
            session.configure(32, 8, 0, 4)
            free, _total = torch.cuda.mem_get_info()
            # leave ~12 GiB free: 5/8 of it holds 48 of the 40 MiB arenas, 129 teams then take 3 sweeps
            ballast = torch.empty(max(0, free - (12 << 30)), dtype=torch.uint8, device="cuda")
        dec(out, wide, prog)
        shape = session.last_decode_shape()
        assert tuple(shape) == (32, 8, 4), shape
    finally:
        del ballast
        session.configure(0, 0, 0)
    del wide
    whole = frames // 32 * 32
    blocks = out[:, :whole].view(N // 2, whole // 32, 32)
    assert bool((blocks == blocks[:, :1, :]).all()), mode
    assert "%016x" % fnv1a64(out[:, :32].contiguous().cpu().numpy()) == k["out_fnv"], mode
    del out, blocks
    torch.cuda.empty_cache()


@pytest.mark.parametrize("fpw,rl,waves,team", [(128, 6, 16, 1), (128, 6, 8, 1), (4, 11, 8, 1), (32, 8, 24, 1), (32, 8, 24, 4), (64, 7, 8, 4)])
def test_persistent_sweeps_vs_oracle(session, oracle, fpw, rl, waves, team):
    """Few resident wavefronts, many groups each: the `base += nb` sweep loop with stale LDS /
    scratch planes and the erasure flags of the previous group, on inputs full of erasures."""
    session.configure(fpw, rl, waves, team)
    try:
        for M, frames in ((8, 40 * fpw), (10, 37 * fpw + 5), (12, 5 * waves * fpw + 3)):
            N = 1 << M
            fro = _code(oracle, M)
            prog = oracle.compile(fro, M)
            cw = oracle.encode(pm1_message(M, N // 2, frames), fro, M, 0)
            for fam in ("small", "uniform"):
                llr = family_llr(fam, 3 * M + frames, cw)
                assert (_decode(session, llr, prog, N // 2) == oracle.decode(llr, prog)).all(), (fpw, rl, waves, team, M, fam)
    finally:
        session.configure(0, 0, 0)


def test_inputs_produced_on_another_stream(session, oracle):
    """include/polar_hip.h, stream ordering: the producer's stream becomes the handle's stream, so a
    decode queued right behind the kernels that write its input needs no host synchronisation."""
    import polar_amd

    M, frames = 12, 4096
    N = 1 << M
    fro = _code(oracle, M)
    prog = oracle.compile(fro, M)
    cw = oracle.encode(pm1_message(5, N // 2, frames), fro, M, 0)
    llr = family_llr("awgn-1", 9, cw)
    exp = oracle.decode(llr, prog)
    src = torch.from_numpy(llr).cuda()
    junk = torch.zeros((N, frames), dtype=torch.int8, device="cuda")
    out = torch.zeros((N // 2, frames), dtype=torch.int8, device="cuda")
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    dec = polar_amd.PolarDecoder(session)
    with torch.cuda.stream(side):
        buf = junk
        for _ in range(200):  # keep the stream busy so that the input lands late
            buf = buf + 1
        buf = buf * 0 + src
        dec(out, buf, prog)
    side.synchronize()
    assert (out.cpu().numpy() == exp).all()
    torch.cuda.synchronize()
    dec(out, src, prog)  # back on the default stream
    assert (out.cpu().numpy() == exp).all()


def test_program_ids_are_deduplicated(session, oracle):
    import polar_amd

    fro = _code(oracle, 9)
    prog = oracle.compile(fro, 9)
    a = polar_amd.lib.polar_hip_load_program(session._h, prog.ctypes.data, prog.size)
    b = polar_amd.lib.polar_hip_load_program(session._h, prog.ctypes.data, prog.size)
    assert a >= 0 and a == b


def test_decoder_every_opcode(session, oracle):
    rng = np.random.default_rng(11)
    seen = set()
    for M in range(1, 10):
        N = 1 << M
        for trial in range(12):
            b = 1 << int(rng.integers(0, M + 1))
            fr = np.repeat((rng.random(N // b) < rng.random()).astype(np.uint8), b)
            for i in range(0, N, 2):
                if fr[i] == 0 and fr[i + 1] == 1:
                    fr[i], fr[i + 1] = 1, 0
            prog = oracle.compile(fr, M)
            seen.update(prog[1:-1].tolist())
            K = int((fr == 0).sum())
            llr = rand_bytes(M * 100 + trial, (N, 36)).view(np.int8)
            assert (_decode(session, llr, prog, K) == oracle.decode(llr, prog)).all(), (M, trial)
    assert seen == set(range(10))


def test_empty_and_device_buffers(session, oracle):
    import polar_amd

    M = 10
    N = 1 << M
    fro = _code(oracle, M)
    prog = oracle.compile(fro, M)
    dec = polar_amd.PolarDecoder(session)
    dec(np.zeros((N // 2, 0), np.int8), np.zeros((N, 0), np.int8), prog)  # empty batch is a no-op
    cw = oracle.encode(pm1_message(1, N // 2, 512), fro, M, 0)
    llr = family_llr("awgn-1", 3, cw)
    d_llr = torch.from_numpy(llr).cuda()
    d_msg = torch.zeros((N // 2, 512), dtype=torch.int8, device="cuda")
    dec(d_msg, d_llr, prog)
    assert (d_msg.cpu().numpy() == oracle.decode(llr, prog)).all()
    assert (d_llr.cpu().numpy() == llr).all()  # the codeword is not modified (polar_decoder.hh:136-137)
    with pytest.raises(polar_amd.PolarError):
        dec(d_msg, d_llr, np.array([10, 0, 255], np.uint8))


def test_encoders(session, oracle, golden):
    import polar_amd

    meta, arr = golden
    for k in meta["encoder_kats"]:
        M = k["M"]
        N = 1 << M
        fro = _code(oracle, M)
        msg = pm1_message(k["msg_seed"], N // 2, k["frames"])
        cw = np.zeros((N, k["frames"]), np.int8)
        enc = (polar_amd.PolarSysEnc if k["systematic"] else polar_amd.PolarEncoder)(M, session)
        enc(cw, msg, fro)
        assert "%016x" % fnv1a64(cw) == k["cw_fnv"], k
    fro = _code(oracle, 8)
    out = np.zeros((256, arr["enc_ternary_in"].shape[1]), np.int8)
    polar_amd.PolarEncoder(8, session)(out, np.ascontiguousarray(arr["enc_ternary_in"]), fro)
    assert (out == arr["enc_ternary_out"]).all()
    # wide (multiple-of-4 frames) path and other rates against the oracle
    for M, K, frames in ((6, 20, 64), (11, 700, 128), (14, 8192, 32)):
        fro = _code(oracle, M, K=K)
        msg = pm1_message(M, K, frames)
        for sysf in (0, 1):
            cw = np.zeros((1 << M, frames), np.int8)
            (polar_amd.PolarSysEnc if sysf else polar_amd.PolarEncoder)(M, session)(cw, msg, fro)
            assert (cw == oracle.encode(msg, fro, M, sysf)).all(), (M, sysf)
    # whole 32-frame columns go through the bit-packed passes: ternary symbols, codes of one
    # pass and of several, K = 0 and K = N, and the byte path beside it (frames not a multiple of 32)
    rng = np.random.default_rng(5)
    for M in (1, 2, 5, 6, 10, 11, 13):
        for K in sorted({0, 1, (1 << M) // 3, 1 << M}):
            fro = _code(oracle, M, K=K)
            for frames in (64, 36):
                msg = rng.integers(-1, 2, (max(K, 1), frames)).astype(np.int8)[:K]
                for sysf in (0, 1):
                    cw = np.zeros((1 << M, frames), np.int8)
                    (polar_amd.PolarSysEnc if sysf else polar_amd.PolarEncoder)(M, session)(cw, msg, fro)
                    assert (cw == oracle.encode(msg, fro, M, sysf)).all(), (M, K, frames, sysf)


def test_host_buffers_pipelined(session, oracle, monkeypatch):
    """polar_hip_decode with several chunks' worth of frames in host buffers takes the chunked
    paths (copy-in, decode and copy-out overlapped on three streams): same bytes as the oracle,
    ragged last chunk included, for torch-pinned buffers, the library's own page-locked ones and
    pageable numpy arrays, with and without message rows."""
    import torch
    import polar_amd

    monkeypatch.setenv("POLAR_HIP_HOST_CHUNK", "1024")
    M, frames = 8, 3 * 1024 + 165
    N = 1 << M
    for K in (N // 2, 0):
        fro = _code(oracle, M, K=K)
        prog = oracle.compile(fro, M)
        llr = family_llr("uniform", 31, np.ones((N, frames), np.int8))
        exp = oracle.decode(llr, prog)
        t_llr = torch.from_numpy(llr).pin_memory()
        t_out = torch.zeros((max(K, 1), frames), dtype=torch.int8).pin_memory()
        polar_amd.PolarDecoder(session)(t_out.numpy()[:K], t_llr.numpy(), prog)
        assert (t_out.numpy()[:K] == exp).all(), K
        h_llr, h_out = polar_amd.HostBuffer((N, frames)), polar_amd.HostBuffer((max(K, 1), frames))
        h_llr.array[:] = llr
        h_out.array[:] = 0
        polar_amd.PolarDecoder(session)(h_out.array[:K], h_llr.array, prog)
        assert (h_out.array[:K] == exp).all(), K
        h_llr.close()
        h_out.close()
        out = np.zeros((max(K, 1), frames), np.int8)  # pageable: gathered into page-locked slots by host threads
        polar_amd.PolarDecoder(session)(out[:K], llr, prog)
        assert (out[:K] == exp).all(), K


def test_systematic_chain(session, oracle):
    """testbench.cc:129-183 on the device: sysenc -> (noise) -> decode -> re-encode -> gather."""
    import polar_amd

    for M, frames in ((10, 100), (12, 64)):
        N = 1 << M
        fro = _code(oracle, M)
        prog = oracle.compile(fro, M)
        msg = pm1_message(9, N // 2, frames)
        cw = np.zeros((N, frames), np.int8)
        polar_amd.PolarSysEnc(M, session)(cw, msg, fro)
        assert (cw[fro == 0] == msg).all()  # testbench.cc:143-146
        for fam in ("awgn+1", "uniform"):
            llr = family_llr(fam, 21, cw)
            dec = _decode(session, llr, prog, N // 2).copy()
            polar_amd.PolarEncoder(M, session).extract_systematic(dec, fro)
            exp = oracle.systematic_extract(oracle.decode(llr, prog), fro, M)
            assert (dec == exp).all(), (M, fam)
            if fam == "awgn+1":
                assert (dec == msg).all()


def test_packed_systematic_extract(session, oracle):
    """the fused re-encode + gather with the bit-packed result; unpacking the planes
    must give oracle.systematic_extract bit for bit, erasures included, ragged and aligned frame
    counts, host and device buffers, codes below and above one pass of the transform."""
    import polar_amd

    for M, frames in ((3, 5), (6, 64), (7, 33), (10, 100), (12, 64), (13, 96), (14, 32)):
        N = 1 << M
        fro = _code(oracle, M)
        K = int((fro == 0).sum())
        prog = oracle.compile(fro, M)
        cw = oracle.encode(pm1_message(M + 40, K, frames), fro, M, 1)
        for fam in ("awgn-1", "uniform"):
            llr = family_llr(fam, 17 + M, cw)
            dec = oracle.decode(llr, prog)  # u-domain decisions, with erasures for "uniform"
            exp = oracle.systematic_extract(dec, fro, M)
            cols = (frames + 31) // 32
            enc = polar_amd.PolarEncoder(M, session)
            inplace = dec.copy()
            enc.extract_systematic(inplace, fro)
            assert (inplace == exp).all(), (M, frames, fam)
            sign = np.full((K, 4 * cols), 0xAA, np.uint8)
            eras = np.full((K, 4 * cols), 0x55, np.uint8)
            keep = dec.copy()
            enc.extract_systematic_packed(sign, eras, keep, fro)
            assert (keep == dec).all()
            bits = np.unpackbits(sign, axis=1, bitorder="little")[:, :frames]
            ebits = np.unpackbits(eras, axis=1, bitorder="little")[:, :frames]
            assert (bits == (exp < 0)).all() and (ebits == (exp == 0)).all(), (M, frames, fam)
            assert not np.unpackbits(eras, axis=1, bitorder="little")[:, frames:].any()  # padding bits are clear
            back = np.zeros_like(exp)
            session.unpack_symbols(back, sign, eras)
            assert (back == exp).all()
            # device buffers
            d_dec = torch.from_numpy(dec).cuda()
            d_sign = torch.zeros((K, 4 * cols), dtype=torch.uint8, device="cuda")
            d_eras = torch.zeros_like(d_sign)
            enc.extract_systematic_packed(d_sign, d_eras, d_dec, fro)
            d_back = torch.zeros((K, frames), dtype=torch.int8, device="cuda")
            session.unpack_symbols(d_back, d_sign, d_eras)
            assert (d_back.cpu().numpy() == exp).all()
            enc.extract_systematic(d_dec, fro)
            assert (d_dec.cpu().numpy() == exp).all()
            if fam == "awgn-1" and M >= 10:
                sign_only = np.zeros((K, 4 * cols), np.uint8)
                enc.extract_systematic_packed(sign_only, None, keep, fro)
                assert (sign_only == sign).all()


def test_full_size_properties(session, oracle):
    """N = 2^16 at batch sizes the oracle cannot follow: noise-free and mildly noisy systematic
    codewords must decode to the transmitted message; a checksum over a subset is compared
    with the oracle."""
    import polar_amd

    M, frames = 16, 2048
    N = 1 << M
    fro = _code(oracle, M)
    prog = oracle.compile(fro, M)
    g = torch.Generator(device="cuda").manual_seed(5)
    msg = (1 - 2 * torch.randint(0, 2, (N // 2, frames), generator=g, device="cuda", dtype=torch.int8)).contiguous()
    cw = torch.zeros((N, frames), dtype=torch.int8, device="cuda")
    polar_amd.PolarSysEnc(M, session)(cw, msg, fro)
    assert bool((cw[torch.from_numpy(fro == 0).cuda()] == msg).all())
    sigma = float(np.sqrt(1.0 / (2.0 * 10.0 ** (0.0 / 10.0))))  # Es/N0 = 0 dB: far above the waterfall
    noisy = cw.to(torch.float32) + sigma * torch.randn(cw.shape, generator=g, device="cuda")
    llr = torch.clamp(torch.round(noisy * (2.0 / sigma**2)), -128, 127).to(torch.int8).contiguous()
    out = torch.zeros((N // 2, frames), dtype=torch.int8, device="cuda")
    polar_amd.PolarDecoder(session)(out, llr, prog)
    sub = llr[:, :48].cpu().numpy().copy()
    assert (out[:, :48].cpu().numpy() == oracle.decode(sub, prog)).all()
    polar_amd.PolarEncoder(M, session).extract_systematic(out, fro)
    assert bool((out == msg).all())


def test_counters(session, oracle):
    """testbench.cc:185-192 on the device against the oracle's counters, host and device buffers."""
    for n_code, n_mesg, seed in ((4096 * 7, 2048 * 7, 1), (1000 * 3 + 5, 500 * 3 + 1, 2), (16, 0, 3)):
        noisy = (rand_bytes(seed, (n_code,)) % 9).astype(np.int8) - 4
        orig = pm1_message(seed + 10, 1, n_code)[0]
        decoded = ((rand_bytes(seed + 20, (max(n_mesg, 1),)) % 3).astype(np.int8) - 1)[:n_mesg]
        message = pm1_message(seed + 30, 1, max(n_mesg, 1))[0][:n_mesg]
        exp = oracle.count(noisy, orig, decoded, message)
        c = np.array([5, 6, 7, 8], np.int64)
        session.count(c, noisy, orig, decoded if n_mesg else None, message if n_mesg else None)
        assert (c - [5, 6, 7, 8]).tolist() == exp.tolist()
        d = torch.zeros(4, dtype=torch.int64, device="cuda")
        tn, to = torch.from_numpy(noisy).cuda(), torch.from_numpy(orig.copy()).cuda()
        if n_mesg:
            session.count(d, tn, to, torch.from_numpy(decoded.copy()).cuda(), torch.from_numpy(message.copy()).cuda())
        else:
            session.count(d, tn, to, None, None)
        assert d.cpu().tolist() == exp.tolist()


def test_channel_and_message_vs_oracle(session, oracle):
    """polar_hip_random_message / polar_hip_channel against the CPU restatement. The message bits
    are integer work: exact. The channel goes through double log / sqrt / sin / cos on both sides;
    a last-place difference there can move a sample across a rounding boundary of quant(), so
    the stated tolerance is: at most 1 in 10^5 LLRs differing, and by no more than 1."""
    rows = 96
    for frames, first in ((257, 0), (64, 4), (130, 1001), (3, 2)):
        got = np.zeros((rows, frames), np.int8)
        session.random_message(got, 11, first)
        assert (got == oracle.random_message(rows, frames, 11, first)).all(), (frames, first)
        cw = got  # any +-1 array
        for snr in (-3.0, -1.0, 4.0, 30.0):
            llr = np.zeros_like(cw)
            session.channel(llr, cw, snr, 5, first)
            exp = oracle.channel(cw, snr, 5, first)
            diff = llr.astype(int) - exp.astype(int)
            assert np.abs(diff).max() <= 1 and (diff != 0).mean() <= 1e-5, (frames, first, snr)
    # device buffers, sharded generation equals the whole batch
    d_cw = torch.from_numpy(oracle.random_message(rows, 4096, 3)).cuda()
    whole = torch.zeros_like(d_cw)
    session.channel(whole, d_cw, -1.0, 8, 0)
    parts = []
    for a, b in ((0, 1000), (1000, 1003), (1003, 4096)):
        part = torch.zeros((rows, b - a), dtype=torch.int8, device="cuda")
        session.channel(part, d_cw[:, a:b].contiguous(), -1.0, 8, a)
        parts.append(part)
    assert bool((torch.cat(parts, dim=1) == whole).all())


def test_channel_screening_is_exact(session, monkeypatch):
    """The channel kernel computes every quad in single precision first and keeps that result
    unless a value lies within the error bound of a rounding boundary, where it redoes the quad
    in double: over 2^27 samples per SNR the LLRs must equal those of the double-only kernel."""
    rows, frames = 2048, 1 << 16
    g = torch.Generator(device="cuda").manual_seed(4)
    cw = (1 - 2 * torch.randint(0, 2, (rows, frames), generator=g, device="cuda", dtype=torch.int8))
    cw[::7, ::5] = 0  # erased code symbols
    for snr in (-3.0, -1.0, 0.5, 5.0, 12.0, 30.0):
        fast = torch.zeros_like(cw)
        session.channel(fast, cw, snr, 21, 4096)
        monkeypatch.setenv("POLAR_HIP_CHANNEL_F64", "1")
        slow = torch.zeros_like(cw)
        session.channel(slow, cw, snr, 21, 4096)
        monkeypatch.delenv("POLAR_HIP_CHANNEL_F64")
        assert bool((fast == slow).all()), snr


def test_simulate_matches_the_step_by_step_chain(session, oracle):
    """polar_hip_simulate (testbench.cc:125-193 on the device) against the same chain assembled
    from the oracle's pieces on the device-generated LLRs, both encoders, sharded and whole."""
    import polar_amd

    for M, frames, systematic, snr in ((8, 300, 0, 1.0), (10, 1100, 1, -1.0), (12, 260, 1, -3.0)):
        N = 1 << M
        fro = _code(oracle, M)
        prog = oracle.compile(fro, M)
        K = N // 2
        seed, first = 77 + M, 40
        got = np.zeros(6, np.int64)
        session.simulate(got, prog, fro, M, systematic, snr, seed, first, frames)
        msg = oracle.random_message(K, frames, seed, first)
        cw = oracle.encode(msg, fro, M, systematic)
        llr = np.zeros_like(cw)
        session.channel(llr, cw, snr, seed, first)  # the device's own LLRs (see the tolerance above)
        dec = oracle.decode(llr, prog)
        if systematic:
            dec = oracle.systematic_extract(dec, fro, M)
        exp = oracle.count(llr, cw, dec, msg)
        assert got[:4].tolist() == list(exp), (M, got, exp)
        assert got[4] == frames and got[5] == frames * K
        # two shards into a device counter array add up to the same totals
        d = torch.zeros(6, dtype=torch.int64, device="cuda")
        session.simulate(d, prog, fro, M, systematic, snr, seed, first, 100)
        session.simulate(d, prog, fro, M, systematic, snr, seed, first + 100, frames - 100)
        assert d.cpu().numpy().tolist() == got.tolist()


def test_float_decoder_vs_oracle(session, oracle):
    """polar_hip_decode_f32 against the oracle's restatement of the reference's float
    helper (itself pinned bit for bit to PolarDecoder<SIMD<float,8>,M>, tests/test_oracle_vs_ref.py).
    Bit-exact, signed zeros included (compared as raw bytes): the float path uses exact products
    with +-1 / 0 and IEEE additions only, so no tolerance is needed."""
    import polar_amd

    dec = polar_amd.PolarDecoder(session)
    for M, frames in ((3, 5), (6, 64), (8, 100), (10, 130), (12, 64), (14, 70)):
        N = 1 << M
        for code in ("const0", "freezer"):
            fro = _code(oracle, M, code)
            K = int((fro == 0).sum())
            prog = oracle.compile(fro, M)
            cw = oracle.encode(pm1_message(M, K, frames), fro, M, 0)
            for fam in ("awgn-1", "uniform", "small"):
                llr = family_llr(fam, 7 * M + 2, cw).astype(np.float32)
                exp = oracle.decode_f32(llr, prog)
                got = np.full((K, frames), 9.0, np.float32)
                dec(got, llr, prog)
                assert got.tobytes() == exp.tobytes(), (M, code, fam)
            frac = ((rand_bytes(M, (N, frames)).astype(np.float32) - 127.5) / 16.0 * cw).astype(np.float32)
            d_in = torch.from_numpy(frac).cuda()
            d_out = torch.zeros((K, frames), dtype=torch.float32, device="cuda")
            dec(d_out, d_in, prog)
            assert d_out.cpu().numpy().tobytes() == oracle.decode_f32(frac, prog).tobytes(), (M, code, "fractional")
    # above the waterfall the float and the int8 decoder both return the transmitted message
    M, frames = 11, 64
    fro = _code(oracle, M)
    prog = oracle.compile(fro, M)
    msg = pm1_message(4, 1 << (M - 1), frames)
    llr = family_llr("awgn+1", 77, oracle.encode(msg, fro, M, 0))
    f = np.zeros((1 << (M - 1), frames), np.float32)
    dec(f, llr.astype(np.float32), prog)
    assert (f == msg).all() and (_decode(session, llr, prog, 1 << (M - 1)) == msg).all()


def test_float_bench_path(session, oracle):
    """SURVEY 8 f3 through the functor interface (`typedef float code_type`, testbench.cc:49-53):
    float encoders, float channel and polar_hip_simulate_f32 against the oracle's pieces. Symbols
    are exact in float; the channel passes through double log / sqrt / sin / cos on both sides and
    is then narrowed to float, so its stated tolerance is: at most 1 LLR in 10^4 differing, and by
    no more than 2 float ulps. The counters are exact on the device's own LLRs."""
    import polar_amd

    for M, frames, systematic, snr in ((8, 200, 0, 1.0), (10, 333, 1, -1.0), (11, 70, 1, -3.0)):
        N = 1 << M
        K = N // 2
        fro = _code(oracle, M)
        prog = oracle.compile(fro, M)
        seed, first = 31 + M, 12
        msg = oracle.random_message(K, frames, seed, first)
        msgf = msg.astype(np.float32)
        for sysm, enc in ((0, polar_amd.PolarEncoder(M, session)), (1, polar_amd.PolarSysEnc(M, session))):
            exp = oracle.encode(msg, fro, M, sysm).astype(np.float32)
            cwf = np.full((N, frames), 9.0, np.float32)
            enc(cwf, msgf, fro)
            assert (cwf == exp).all(), (M, sysm)
            d = torch.zeros((N, frames), dtype=torch.float32, device="cuda")
            enc(d, torch.from_numpy(msgf).cuda(), fro)
            assert (d.cpu().numpy() == exp).all(), (M, sysm, "device")
            with pytest.raises(polar_amd.PolarError):
                enc(cwf, msgf * 0.5, fro)  # not symbols: the int8 kernels cannot carry them
        cw = oracle.encode(msg, fro, M, systematic)
        llr = np.zeros((N, frames), np.float32)
        session.channel(llr, cw.astype(np.float32), snr, seed, first)
        exp = oracle.channel_f32(cw, snr, seed, first)
        off = llr != exp
        assert off.mean() <= 1e-4 and (np.abs(llr - exp)[off] <= 2 * np.spacing(np.abs(exp[off]))).all(), (M, int(off.sum()))
        got = np.zeros(6, np.int64)
        session.simulate(got, prog, fro, M, systematic, snr, seed, first, frames, code_type="float")
        dec = oracle.decode_f32(llr, prog).astype(np.int8)
        if systematic:
            dec = oracle.systematic_extract(dec, fro, M)
        want = [int((llr * cw < 0).sum()), int((llr == 0).sum()), int((dec.astype(int) * msg <= 0).sum()), int((dec == 0).sum())]
        assert got[:4].tolist() == want, (M, got, want)
        assert got[4] == frames and got[5] == frames * K
        d = torch.zeros(6, dtype=torch.int64, device="cuda")
        session.simulate(d, prog, fro, M, systematic, snr, seed, first, 64, code_type="float")
        session.simulate(d, prog, fro, M, systematic, snr, seed, first + 64, frames - 64, code_type="float")
        assert d.cpu().numpy().tolist() == got.tolist()
        # the two code types meet the same frames: well above the waterfall both decode everything
        a, b = np.zeros(6, np.int64), np.zeros(6, np.int64)
        session.simulate(a, prog, fro, M, systematic, 6.0, seed, first, frames)
        session.simulate(b, prog, fro, M, systematic, 6.0, seed, first, frames, code_type="float")
        assert a[2] == 0 and b[2] == 0 and a[4:].tolist() == b[4:].tolist()


def test_code_cache_and_sweep_harness(session, oracle):
    """the handle's (level, K, p) cache returns the reference's construction and one
    resident program per key; the sweep harness on top of polar_hip_simulate prints the reference's
    four columns and finds the waterfall where ber_log.png (README.md:24) has it for N = 2^12..2^14:
    errors at -3 dB, none at +1 dB. (The reference's own random stream cannot be reproduced, so the
    curve itself is unpinned; every counter behind it is pinned by test_simulate_*.)"""
    import io

    from polar_amd.sweep import ber_sweep, design_probability, design_snr

    M = 12
    N = 1 << M
    p = design_probability(design_snr())
    fro, prog, amb = session.code(M, N // 2, p)
    ofro, oamb = oracle.const0(M, N // 2, p)
    assert (fro == ofro).all() and amb == bool(oamb) and (prog == oracle.compile(ofro, M)).all()
    pid = session.load_program(prog)[0]
    fro2, prog2, _ = session.code(M, N // 2, p)
    assert session.load_program(prog2)[0] == pid and (fro2 == fro).all()
    fro3, prog3, _ = session.code(M, N // 2, design_probability(0.0))
    assert session.load_program(prog3)[0] != pid and (fro3 != fro).any()
    out = io.StringIO()
    rows = ber_sweep(session, M, systematic=True, chunk=2048, max_frames=4096, out=out)
    by = {round(r[0], 1): r for r in rows}
    assert by[-3.0][1] > 1e-2 and rows[-1][1] == 0.0 and -1.5 <= rows[-4][0] <= 1.0
    assert all(len(x.split()) == 4 for x in out.getvalue().splitlines() if not x.startswith("#"))
    rows_e = ber_sweep(session, M, systematic=False, chunk=2048, max_frames=2048, per_snr_design=True, out=io.StringIO())
    assert rows_e[0][1] > 1e-2 and rows_e[-1][1] == 0.0


def test_mixed_sizes_in_one_launch_set(session, oracle):
    """BASELINE config E: codes of N = 2^10 .. 2^16, each with its own design SNR (its own frozen
    set and program), resident together and decoded back to back on one stream without a host
    synchronisation in between."""
    import polar_amd

    dec = polar_amd.PolarDecoder(session)
    stream = torch.cuda.Stream()
    jobs = []
    for k, M in enumerate(range(10, 17)):
        N = 1 << M
        p = float(np.exp(-(10.0 ** ((-2.0 + 0.5 * k) / 10.0))))  # design point moves with the code, testbench.cc:74-88
        fro, _ = oracle.const0(M, N // 2, p)
        prog = oracle.compile(fro, M)
        frames = 24 + 13 * k
        msg = oracle.random_message(N // 2, frames, 100 + M)
        llr = oracle.channel(oracle.encode(msg, fro, M, 1), 0.0, 200 + M)
        jobs.append((prog, llr, torch.from_numpy(llr).cuda(), torch.zeros((N // 2, frames), dtype=torch.int8, device="cuda")))
    torch.cuda.synchronize()
    for prog, _llr, d_llr, d_out in jobs:
        dec.launch(d_out, d_llr, prog, stream.cuda_stream)
    stream.synchronize()
    for prog, llr, _d, d_out in jobs:
        assert (d_out.cpu().numpy() == oracle.decode(llr, prog)).all(), int(prog[0])
    # the same set through polar_hip_decode_set: the library spreads the jobs over its four lanes; twice, the
    # second time with the jobs in another order and one of them empty
    for rnd in range(2):
        for _p, _l, _d, d_out in jobs:
            d_out.fill_(9)
        sel = jobs if rnd == 0 else jobs[::-1]
        set_jobs = [(d_out, d_llr, prog) for prog, _llr, d_llr, d_out in sel]
        if rnd:
            N = 1 << 10
            set_jobs.append((torch.zeros((N // 2, 0), dtype=torch.int8, device="cuda"), torch.zeros((N, 0), dtype=torch.int8, device="cuda"), jobs[0][0]))
        dec.decode_set(set_jobs)
        for prog, llr, _d, d_out in jobs:
            assert (d_out.cpu().numpy() == oracle.decode(llr, prog)).all(), (rnd, int(prog[0]))
    # whole teams (multiples of 128 frames) of codes that do not fill the device share ONE kernel launch; a batch
    # that does fill it (N = 2^10, 2^19 frames, checked by tiling) is launched on its own inside the same call
    shared = []
    for k, M in enumerate(range(10, 15)):
        N = 1 << M
        fro, _ = oracle.const0(M, N // 2, float(np.exp(-(10.0 ** ((-1.0 + 0.3 * k) / 10.0)))))
        prog = oracle.compile(fro, M)
        frames = 128 * (1 + (k * 2) % 5)
        msg = oracle.random_message(N // 2, frames, 300 + M)
        llr = oracle.channel(oracle.encode(msg, fro, M, 1), -1.0, 400 + M)
        shared.append((prog, llr, torch.from_numpy(llr).cuda(), torch.full((N // 2, frames), 9, dtype=torch.int8, device="cuda")))
    prog0, llr0 = shared[0][0], shared[0][1][:, :32]
    big_in = torch.from_numpy(np.ascontiguousarray(llr0)).cuda().repeat(1, (1 << 19) // 32).contiguous()
    big_out = torch.full((512, 1 << 19), 9, dtype=torch.int8, device="cuda")
    dec.decode_set([(d_out, d_llr, prog) for prog, _llr, d_llr, d_out in shared] + [(big_out, big_in, prog0)])
    for prog, llr, _d, d_out in shared:
        assert (d_out.cpu().numpy() == oracle.decode(llr, prog)).all(), ("shared launch", int(prog[0]))
    exp0 = torch.from_numpy(oracle.decode(np.ascontiguousarray(llr0), prog0)).cuda()
    assert bool((big_out.view(512, -1, 32) == exp0.view(512, 1, 32)).all())
    del big_in, big_out
    # the same shared launch with rows further apart than their length (polar_hip_job's pitches): input and output
    # as column ranges of wider tensors; nothing outside the ranges is written
    pitched = []
    for k, (prog, llr, d_llr, _o) in enumerate(shared):
        frames = llr.shape[1]
        wide_in = torch.full((llr.shape[0], frames + 16 * (k + 1)), 77, dtype=torch.int8, device="cuda")
        wide_in[:, 16:16 + frames] = d_llr
        wide_out = torch.full((llr.shape[0] // 2, frames + 4352), 9, dtype=torch.int8, device="cuda")
        pitched.append((prog, llr, wide_in, wide_out))
    dec.decode_set([(wo[:, :llr.shape[1]], wi[:, 16:16 + llr.shape[1]], prog) for prog, llr, wi, wo in pitched])
    for prog, llr, _wi, wo in pitched:
        assert (wo[:, :llr.shape[1]].cpu().numpy() == oracle.decode(llr, prog)).all(), ("pitched shared launch", int(prog[0]))
        assert bool((wo[:, llr.shape[1]:] == 9).all())
    with pytest.raises(ValueError):
        dec.decode_set([(np.zeros((512, 4), np.int8), np.zeros((1024, 4), np.int8), jobs[0][0])])  # host buffers are refused


def test_cpp_dropin_header():
    """include/polar_amd/polar.hh: the reference bench's call sequence with TYPE = GPU<int8_t, W>
    (tests/cpp/dropin_test.cc, built by __graft_entry__.build())."""
    import os
    import subprocess

    from conftest import ROOT

    exe = os.path.join(ROOT, "tests", "cpp", "dropin_test")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "dropin ok" in r.stdout, r.stdout + r.stderr
    # the same functors on buffers the caller keeps in device memory (tests/cpp/dropin_device_test.cc)
    exe = os.path.join(ROOT, "tests", "cpp", "dropin_device_test")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "dropin device ok" in r.stdout, r.stdout + r.stderr
    # the bench's float build, TYPE = GPU<float, W> (tests/cpp/dropin_float_test.cc)
    exe = os.path.join(ROOT, "tests", "cpp", "dropin_float_test")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "dropin float ok" in r.stdout, r.stdout + r.stderr


def test_cpp_caller_shards_with_rccl():
    """SURVEY 8 e for a C / C++ caller: polar_hip_simulate into device counters per shard, then
    polar_hip_allreduce_counters over the caller's RCCL communicator (tests/cpp/shard_counters_test.cc,
    built by __graft_entry__.build()); a world of one rank on this one-GPU box: two shards add up to
    the whole batch, the all-reduce leaves the sums in place, misuse is refused."""
    import os
    import subprocess

    from conftest import ROOT

    exe = os.path.join(ROOT, "tests", "cpp", "shard_counters_test")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0 and "shard counters ok" in r.stdout, r.stdout + r.stderr


def test_bench_joins_the_rccl_group_as_a_world_of_one():
    """SURVEY 8 e: the N > 1 path of bench.py (init_process_group("nccl"), barriers, counter all-reduce,
    max-over-ranks time) run on hardware with the only world size a one-GPU box offers. A small batch,
    no CPU baseline: the line must parse and carry the counters."""
    import json
    import os
    import subprocess
    import sys

    from conftest import ROOT

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--config", "A", "--frames", "32768", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0"],
                       capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["roofline"]["frac"] > 0


def test_arena_placement_choice_keeps_results(oracle, golden, capfd, monkeypatch):
    """decode_launch.hpp, placement_step: a handle that decodes the same resident batch repeatedly looks at
    several placements of its scratch arenas (sets the current one aside, allocates another, keeps the
    faster). Every call, on whichever arenas it runs, must return the reference's digest; the log shows
    that placements were compared; another workload in between starts the comparison over and a batch
    with small arenas is left alone."""
    import polar_amd

    meta, _ = golden
    M, frames = 16, 1 << 17  # (64, 7) teams: 2048 arenas of 5 MiB
    N = 1 << M
    fro = _code(oracle, M)
    prog = oracle.compile(fro, M)
    k = [k for k in meta["decoder_digests"] if k["M"] == M][0]
    cw = oracle.encode(pm1_message(k["msg_seed"], N // 2, k["frames"]), fro, M, 1)
    llr = family_llr(k["family"], k["llr_seed"], cw)
    wide = torch.from_numpy(llr).cuda().repeat(1, frames // 32).contiguous()
    small = wide[:, :256].contiguous()
    monkeypatch.setenv("POLAR_HIP_DEBUG", "1")
    sess = polar_amd.Session(0)
    dec = polar_amd.PolarDecoder(sess)
    out = torch.empty((N // 2, frames), dtype=torch.int8, device="cuda")
    out_small = torch.empty((N // 2, 256), dtype=torch.int8, device="cuda")

    def check(o, f):
        blocks = o.view(N // 2, f // 32, 32)
        assert bool((blocks == blocks[:, :1, :]).all())
        assert "%016x" % fnv1a64(o[:, :32].contiguous().cpu().numpy()) == k["out_fnv"]

    for call in range(7):
        out.fill_(55)
        dec(out, wide, prog)
        check(out, frames)
        if call == 2:  # another workload between two calls: small arenas, no comparison, and the big one starts over
            out_small.fill_(55)
            dec(out_small, small, prog)
            check(out_small, 256)
    log = capfd.readouterr().err
    assert log.count("arena placement") >= 4 and "arena placement 1:" in log, log[-2000:]
    sess.close()


@pytest.mark.gpu
@pytest.mark.parametrize("M,frames", [(10, 512 * 2001), (12, 512 * 601)])
def test_relay_launches_keep_results(oracle, monkeypatch, M, frames):
    """decode_launch.hpp, relay launches: a batch of several sweeps goes out as two cohorts of half a grid each on
    two streams, three launches per segment of about four sweeps. The decisions must be those of the single
    launch (POLAR_HIP_RELAY=1) bit for bit, for a batch that does not divide evenly, on repeated calls (the
    arenas keep the last call's contents), and the oracle's on a sample of frames from every launch."""
    import polar_amd

    N = 1 << M
    fro = _code(oracle, M)
    prog = oracle.compile(fro, M)
    sess = polar_amd.Session(0)
    dec = polar_amd.PolarDecoder(sess)
    g = torch.Generator(device="cuda").manual_seed(M)
    llr = torch.randint(-40, 90, (N, frames), dtype=torch.int8, device="cuda", generator=g)
    one = torch.empty((N // 2, frames), dtype=torch.int8, device="cuda")
    monkeypatch.setenv("POLAR_HIP_RELAY", "1")
    dec(one, llr, prog)
    assert sess.last_decode_launches() == 1
    for parts in ("4", "2", "3"):
        monkeypatch.setenv("POLAR_HIP_RELAY", parts)
        for _ in range(2):
            out = torch.full((N // 2, frames), 77, dtype=torch.int8, device="cuda")
            dec(out, llr, prog)
            assert sess.last_decode_launches() > 1 and sess.last_decode_launches() % 3 == 0, (parts, sess.last_decode_launches(), sess.last_decode_shape())
            assert bool((out == one).all())
    monkeypatch.delenv("POLAR_HIP_RELAY")  # the default: relay launches only from 2^34 LLRs on
    dec(torch.empty_like(out), llr, prog)
    assert sess.last_decode_launches() == 1
    cols = np.unique(np.concatenate([np.arange(0, 64), np.linspace(0, frames - 1, 200).astype(np.int64), np.arange(frames - 64, frames)]))
    sample = llr[:, torch.from_numpy(cols).cuda()].cpu().numpy()
    want = oracle.decode(sample, prog)
    assert np.array_equal(out[:, torch.from_numpy(cols).cuda()].cpu().numpy(), want)
    sess.close()


@pytest.mark.gpu
@pytest.mark.parametrize("M,frames,pitches", [(8, 512, (512 + 4352, 1024)), (11, 1024, (1040, 1024 + 16)), (9, 200, (333, 208)), (12, 4096, (0, 8192))])
def test_pitched_rows(oracle, M, frames, pitches):
    """polar_hip_decode_pitched: rows of the caller's buffers further apart than their length (column ranges of
    wider tensors; rows padded to polar_hip_row_pitch). Pitches that are multiples of 16 are used in place, others
    and ragged batches staged; the decisions are the oracle's either way and nothing outside the rows is written."""
    import polar_amd

    N = 1 << M
    fro = _code(oracle, M)
    prog = oracle.compile(fro, M)
    rows = int(N - fro.sum())
    cpitch, mpitch = pitches
    rng = np.random.default_rng(M * 1000 + frames)
    llr = rng.integers(-60, 100, (N, frames), dtype=np.int8)
    want = oracle.decode(llr, prog)
    sess = polar_amd.Session(0)
    dec = polar_amd.PolarDecoder(sess)
    wide_in = torch.full((N, cpitch or frames), 99, dtype=torch.int8, device="cuda")
    cw = wide_in[:, :frames]
    cw.copy_(torch.from_numpy(llr))
    wide_out = torch.full((rows, mpitch or frames), 55, dtype=torch.int8, device="cuda")
    out = wide_out[:, :frames]
    dec(out, cw, prog)
    assert np.array_equal(out.cpu().numpy(), want)
    if mpitch:
        assert bool((wide_out[:, frames:] == 55).all())
    # a sub-batch of a wider tensor, without a copy
    a, b = (frames // 4) // 16 * 16, frames - 16
    wide_out.fill_(55)
    dec(wide_out[:, a:b], wide_in[:, a:b], prog)
    assert np.array_equal(wide_out[:, a:b].cpu().numpy(), want[:, a:b])
    assert bool((wide_out[:, :a] == 55).all()) and bool((wide_out[:, b:] == 55).all())
    assert polar_amd.pitched_empty(4, 1 << 20).stride(0) == (1 << 20) + 4352 and polar_amd.pitched_empty(4, 100).stride(0) == 112
    sess.close()
