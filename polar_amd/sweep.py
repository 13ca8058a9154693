"""The reference bench's outer loop (testbench.cc:74-110, 125, 195-221) over the device's
one-SNR-point simulation (polar_hip_simulate): design the code, sweep Es/N0 upward in steps of
0.1 dB from floor(design - 3), stop after the reference's run of error-free points or at
ceil(design + 5), and print its four columns `SNR BER Mbit/s Eb/N0` per point.

What differs from the reference, on purpose:
  * frames per point: the reference runs at most 320 / SIMD_WIDTH - 1 bundles (288 frames at
    W = 32) or until 1000 uncorrected errors (testbench.cc:125); here a point runs chunks of
    `chunk` frames until the error budget is met or `max_frames` are spent, the counters being
    all-reduced over the ranks between chunks, so the curves reach far lower BER;
  * the frames of a point are a pure function of (seed, SNR index, frame index) -- the reference
    seeds from std::random_device and cannot be reproduced;
  * Mbit/s is the decode kernel alone, as the reference times it (testbench.cc:170-175), summed
    over the ranks.
"""
import math
import sys
import threading
import time

import numpy as np

from .shard import shard_range


def design_snr(rate=0.5):
    """testbench.cc:74-76: the design SNR of a rate-R code, 10 log10(-ln(1 - R))."""
    return 10.0 * math.log10(-math.log(1.0 - rate))


def design_probability(snr_db):
    """testbench.cc:85-87: erasure probability PolarCodeConst0 is fed for a design SNR."""
    return math.exp(-(10.0 ** ((snr_db + 1.59175) / 10.0)))


def host_code(M, K, p):
    """(frozen, program) of PolarCodeConst0(M, K, p) + PolarCompiler on the host, no device handle
    involved (polar_freezer.hh:34-62, polar_compiler.hh:51-58 through the C ABI's host functions)."""
    from . import PolarCodeConst0, PolarCompiler

    N = 1 << M
    fro = np.zeros(N, np.uint8)
    PolarCodeConst0()(fro, M, K, p)
    prog = np.zeros(2 * N + 1, np.uint8)
    n = PolarCompiler()(prog, fro, M)
    return fro, prog[:n].copy()


def snr_points(design):
    """The reference's loop variable: low = floor(design - 3), += 0.1 in double, up to ceil(design + 5)."""
    low, high = math.floor(design - 3.0), math.ceil(design + 5.0)
    snr = low
    while snr <= high:
        yield snr
        snr += 0.1


def ber_sweep(session, M, systematic=True, rate=0.5, seed=42, chunk=1 << 14, max_frames=1 << 20, error_budget=1000,
              per_snr_design=False, dist=None, rank=0, world=1, out=sys.stdout, device=None, clean_points=3):
    """Returns the list of rows (snr, ber, mbit_s, ebn0, frames, counters[6]); rank 0 prints them.
    per_snr_design: redesign the code for every channel SNR (BASELINE config E) through the
    handle's (M, K, p) cache instead of once for the design SNR of the rate."""
    import torch

    N = 1 << M
    K = int(rate * N)  # testbench.cc:75
    design = design_snr(rate)
    fro, prog, amb = session.code(M, K, design_probability(design))
    rows = []
    count = 0
    min_snr, max_mbs = None, 0.0
    if rank == 0:
        print("# Polar(%d, %d) %s, design SNR %.5f dB%s, %d rank(s), <= %d frames per point in chunks of %d" % (
            N, K, "systematic" if systematic else "non-systematic", design, " (tie at the K boundary)" if amb else "", world, max_frames, chunk), file=out)
        print("# SNR BER Mbit/s Eb/N0", file=out)
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
    # per-SNR design: PolarCodeConst0 + PolarCompiler are host work (long double, the
    # reference's own arithmetic). The code of point i+1 is constructed on a host thread while point
    # i decodes on the device; only its upload (polar_hip_load_program, inside simulate) stays on the
    # calling thread, because the handle is not thread-safe. t_wait = what the sweep still waits for.
    t_start, t_wait, t_build = time.perf_counter(), 0.0, [0.0]
    points = list(snr_points(design))
    pending = None

    def build(snr):
        t0 = time.perf_counter()
        r = host_code(M, K, design_probability(snr))
        t_build[0] += time.perf_counter() - t0
        return r

    def prefetch(i):
        if not per_snr_design or i >= len(points):
            return None
        box = {}
        th = threading.Thread(target=lambda: box.update(code=build(points[i])))
        th.start()
        return th, box

    if per_snr_design:
        pending = prefetch(0)
    for idx, snr in enumerate(points):
        if count > clean_points:
            break
        if per_snr_design:
            t0 = time.perf_counter()
            pending[0].join()
            fro, prog = pending[1]["code"]
            t_wait += time.perf_counter() - t0
            pending = prefetch(idx + 1)
        total = torch.zeros(6, dtype=torch.int64, device=dev)
        done, ms, bits = 0, 0.0, 0.0
        while done < max_frames:
            c = min(chunk, max_frames - done)
            a, b = shard_range(c, rank, world)
            part = torch.zeros(6, dtype=torch.int64, device=dev)
            if b > a:
                # frame index space: [SNR point][frame] so that every point has fresh noise and messages
                session.simulate(part, prog, fro, M, systematic, snr, seed, (idx << 40) + done + a, b - a)
                ms += session.last_decode_ms()
                bits += (b - a) * K
            if dist is not None:
                dist.all_reduce(part)
            total += part
            done += c
            if int(total[2].item()) >= error_budget:  # evaluated on the reduced value (testbench.cc:125)
                break
        t = total.cpu().numpy()
        ber = t[2] / float(t[5]) if t[5] else 0.0
        mbs = bits / (ms * 1e-3) / 1e6 * world if ms > 0 else 0.0
        sigma2 = 1.0 / (2.0 * 10.0 ** (snr / 10.0))
        ebn0 = 10.0 * math.log10(1.0 / ((K / float(N)) * 2.0 * sigma2))  # testbench.cc:203-206
        if t[2] == 0:
            min_snr = snr if min_snr is None else min(min_snr, snr)
        else:
            count = 0
        max_mbs = max(max_mbs, mbs)
        rows.append((snr, ber, mbs, ebn0, int(t[4]), t.tolist()))
        if rank == 0:
            print("%g %g %g %g" % (snr, ber, mbs, ebn0), file=out, flush=True)
        count += 1
    if pending is not None:
        pending[0].join()
    if rank == 0:
        print("# QEF at: %s SNR, speed: %g Mb/s." % ("%g" % min_snr if min_snr is not None else "none", max_mbs), file=out, flush=True)
        if per_snr_design:
            wall = time.perf_counter() - t_start
            print("# per-SNR code construction + compilation on the host: %.3f s in all, %.3f s of them waited for (the rest ran beside the previous point's decode); sweep wall %.3f s -> %.2f %% of it" % (
                t_build[0], t_wait, wall, 100.0 * t_wait / wall if wall > 0 else 0.0), file=out, flush=True)
    return rows
