#!/usr/bin/env python3
"""bench.py -- headline benchmark: batched int8 SC decoding of the rate-1/2 N=2^16 polar code.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one resident batch: the SC decode kernel over F
frames per GPU (BASELINE.json configs[1]: N=2^16, K=32768, systematic, 2^20 frames) -- the
region the reference itself times (testbench.cc:170-174). The timed region of the K steps
ends with the bench's error counters (testbench.cc:189-192) over the decoded batch on the
device and, with more than one rank, the RCCL all-reduce of those counters (configs[2],
SURVEY 8d/8e). Frames are independent, so ranks shard them with no other exchange (weak
scaling: F frames per GPU).

Inputs are synthetic and resident in HBM before the timed region: random +-1 messages,
systematic encoding (PolarSysEnc, on device), BPSK/AWGN at Es/N0 = -1 dB, LLR = 2y/sigma^2
quantised to int8 as testbench.cc:112-114,154-165 does -- all on the device through the
library's counter-based message source and channel, so the batch does not depend on how the
frames are sharded over ranks.

Prints ONE JSON line (rank 0). `value` = info bits decoded by all ranks / wall time of the K
steps (max over ranks). `roofline` prices the decode kernel alone (HIP events on its launch
stream) against 8 TB/s with the algorithmic 1.5*N bytes per frame; `cpu_baseline`
times the reference's own AVX2 decoder (oracle/_ref, built from the reference headers) on the
host cores for a bounded sample of the same LLRs.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

M = 16
N = 1 << M
K = N // 2
SNR_DB = -1.0
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=1 << 20, help="frames per GPU")
    ap.add_argument("--fpw", type=int, default=0, help="frames per wavefront (0 = library default)")
    ap.add_argument("--lds-level", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget per thread (0 = skip)")
    return ap.parse_args()


def make_code(polar_amd):
    fro = np.zeros(N, np.uint8)
    polar_amd.PolarCodeConst0()(fro, M, K, polar_amd.testbench_probability())
    prog = np.zeros(2 * N + 1, np.uint8)
    n = polar_amd.PolarCompiler()(prog, fro, M)
    return fro, prog[:n].copy()


def generate(torch, polar_amd, sess, fro, prog, frames, seed, first_frame, dev):
    """Resident batch: llr [N][frames] int8 and the transmitted u-domain message [K][frames].
    Message bits and noise come from the library's counter-based source (polar_hip_random_message,
    polar_hip_channel): global frame g is the same whatever rank or chunk produces it."""
    llr = torch.empty((N, frames), dtype=torch.int8, device=dev)
    u_true = torch.empty((K, frames), dtype=torch.int8, device=dev)
    enc = polar_amd.PolarSysEnc(M, sess)
    dec = polar_amd.PolarDecoder(sess)
    chunk = min(frames, 8192)
    stats = torch.zeros(4, dtype=torch.int64, device=dev)
    for a in range(0, frames, chunk):
        c = min(chunk, frames - a)
        msg = torch.empty((K, c), dtype=torch.int8, device=dev)
        sess.random_message(msg, seed, first_frame + a)
        cw = torch.empty((N, c), dtype=torch.int8, device=dev)
        enc(cw, msg, fro)
        q = torch.empty((N, c), dtype=torch.int8, device=dev)
        sess.channel(q, cw, SNR_DB, seed, first_frame + a)
        sess.count(stats, q, cw, None, None)  # awgn_errors / quantization_erasures of the batch
        llr[:, a:a + c] = q
        # u-domain truth: decoding the noise-free codeword returns the encoder's input
        clean = (cw * 64).contiguous()
        u = torch.empty((K, c), dtype=torch.int8, device=dev)
        dec(u, clean, prog)
        u_true[:, a:a + c] = u
        del cw, q, clean, u, msg
    torch.cuda.synchronize(dev)
    return llr, u_true, stats


def cpu_baseline(llr_sample, prog, seconds):
    """The genuine reference decoder (PolarDecoder<SIMD<int8_t,32>>) on the host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    threads = max(1, min(16, os.cpu_count() or 1))
    if os.path.exists(oracle_lib.REF_SO):
        ref = oracle_lib.Ref()
        W = ref.width
        kind = "reference"

        def timed(bundle, reps):
            return ref.decode_time(bundle, prog, reps)
    else:
        orc = oracle_lib.Oracle()
        W = 32
        kind = "port"

        def timed(bundle, reps):
            return orc.decode_time(bundle, prog, reps)
    bundles = [np.ascontiguousarray(llr_sample[:, t * W:(t + 1) * W]) for t in range(threads)]
    per_call = timed(bundles[0], 20) / 20
    reps = max(1, int(seconds / per_call))
    out = [0.0] * threads

    def work(t):
        out[t] = timed(bundles[t], reps)

    th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    t0 = time.time()
    for x in th:
        x.start()
    for x in th:
        x.join()
    wall = time.time() - t0
    bits = threads * reps * W * K
    return {
        "value": bits / max(out) / 1e9, "unit": "Gbit/s", "cores": threads, "kind": kind,
        "sample": "%d threads x %d calls x %d frames of the same N=2^16 LLR batch, decode only (wall %.1f s, %.1f us per call)" % (threads, reps, W, wall, 1e6 * max(out) / reps),
    }


def main():
    args = parse()
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    import polar_amd

    sess = polar_amd.Session(local)
    if args.fpw:
        sess.configure(args.fpw, args.lds_level, 0)
    fro, prog = make_code(polar_amd)
    frames = args.frames
    llr, u_true, chan = generate(torch, polar_amd, sess, fro, prog, frames, 42, rank * frames, dev)
    out = torch.empty((K, frames), dtype=torch.int8, device=dev)
    counters = torch.zeros(4, dtype=torch.int64, device=dev)
    dec = polar_amd.PolarDecoder(sess)
    kernel_ms = []

    def step():
        dec(out, llr, prog)
        kernel_ms.append(sess.last_decode_ms())

    def tally():
        # testbench.cc:189-192 once per batch, then the one collective of the job
        counters.zero_()
        sess.count(counters, None, None, out, u_true)
        if dist is not None:
            dist.all_reduce(counters)

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    tally()
    kernel_ms.clear()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    tally()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.all_reduce(chan)
    c = counters.cpu().tolist()
    ch = chan.cpu().tolist()
    total_frames = frames * world
    value = total_frames * K * args.steps / elapsed / 1e9
    if rank == 0:
        k_ms = float(np.mean(kernel_ms))
        achieved = frames * 1.5 * N / (k_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            t = json.load(open(tpath))
            if t.get("frames") == frames and t.get("fpw") == args.fpw and t.get("lds_level") == args.lds_level:
                traffic = t.get("bytes_per_launch")
        line = {
            "metric": "decoded info Gbit/s (node) at N=2^16 R=1/2 int8 SC", "value": value, "unit": "Gbit/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int8", "data": "synthetic",
            "config": {
                "workload": "N=2^16 K=32768 systematic Const0 code, %d frames per GPU resident in HBM, BPSK/AWGN Es/N0=%.1f dB, int8 LLRs; step = SC decode of the batch (what testbench.cc:170-174 times); the timed region ends with one error count%s" % (frames, SNR_DB, " + RCCL all-reduce of the counters" if world > 1 else ""),
                "frames_per_gpu": frames, "parallelism": "frames sharded over %d GPU(s)" % world,
                "frames_per_wave": args.fpw or "default", "lds_level": args.lds_level or "default",
            },
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "sc_decode_kernel", "kernel_ms": k_ms,
                         "algorithmic_bytes_per_launch": frames * 1.5 * N},
            "frac_of_hbm_roofline_whole_job": value * 1e9 * 3 / (HBM_PEAK_GBS * 1e9 * world),
            "ber": c[2] / float(total_frames * K), "uncorrected_errors": c[2], "ambiguity_erasures": c[3],
            "awgn_errors": ch[0], "quantization_erasures": ch[1],
        }
        if world == 1 and args.cpu_seconds > 0:
            sample = llr[:, :16 * 32].cpu().numpy()
            line["cpu_baseline"] = cpu_baseline(sample, prog, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
