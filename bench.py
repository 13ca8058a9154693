#!/usr/bin/env python3
"""bench.py -- headline benchmark: batched int8 SC decoding of the rate-1/2 N=2^16 polar code.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F] [--config A|B|D]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no RANK in the environment starts the N ranks itself (one child
process per GPU, spawned before this process touches HIP; RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* are set for them exactly as torch.distributed.run would). Under torchrun the ranks
already exist and `--gpus` must equal WORLD_SIZE.

One "step" = one pass of the hot path over one resident batch: the SC decode kernel over F
frames per GPU (BASELINE.json configs[1]: N=2^16, K=32768, systematic, 2^20 frames) -- the
region the reference itself times (testbench.cc:170-174). The timed region of the K steps
ends with the bench's error counters (testbench.cc:189-192) over the decoded batch on the
device and, with more than one rank, the RCCL all-reduce of those counters (configs[2],
SURVEY 8d/8e). Frames are independent, so ranks shard them with no other exchange (weak
scaling: F frames per GPU). `--include-extract` adds the systematic post-processing of
testbench.cc:177-183 (re-encode + gather, fused on the device) to every step.

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

Other configs of BASELINE.json (`--config`): A = N=2^10 non-systematic (configs[0]), D = N=2^20
systematic (configs[3]), E = mixed N=2^10..2^16 with per-SNR frozen sets (configs[4]); B (default)
is the headline. They change the workload named in `config.workload`, nothing else.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SNR_DB = -1.0
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
CONFIGS = {
    # name: (M, systematic, default frames per GPU)
    "A": (10, 0, 1 << 22),
    "B": (16, 1, 1 << 20),
    "D": (20, 1, 1 << 16),
    # E: BASELINE configs[4] -- one launch set mixing N = 2^10 .. 2^16, every (N, channel SNR) with the frozen
    # set designed for that SNR (testbench.cc:74-88 per point); "frames" = bytes of LLRs per (N, SNR) batch
    "E": (0, 1, 1 << 29),
}
E_LEVELS = range(10, 17)
E_SNRS = (-2.0, -1.5, -1.0, -0.5)


SETTLE_CALLS = 6  # untimed decode calls of the set-up in which the library settles the placement of its arenas and whether to use relay launches


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="B")
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU (0 = the config's default)")
    ap.add_argument("--fpw", type=int, default=0, help="frames per wavefront (0 = library default)")
    ap.add_argument("--lds-level", type=int, default=0)
    ap.add_argument("--include-extract", action="store_true", help="time re-encode + gather (testbench.cc:177-183) with every decode")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget per thread (0 = skip)")
    ap.add_argument("--plumbing-only", action="store_true", help=argparse.SUPPRESS)  # CPU/gloo test of the rank plumbing
    ap.add_argument("--rank-timeout", type=float, default=1500.0, help="seconds the rank launcher waits for its children")
    ap.add_argument("--init-timeout", type=float, default=180.0, help="seconds a rank waits in the process-group rendezvous")
    ap.add_argument("--streams", type=int, default=0, help="config E: 0 = the library's polar_hip_decode_set (four internal lanes); k > 0 = k handles / HIP streams managed here (1 = back to back; experiments)")
    ap.add_argument("--lane-waves", type=int, default=0, help="config E: resident wavefronts per handle (0 = an equal share of the device)")
    ap.add_argument("--contiguous-llr", action="store_true", help="LLR rows exactly `frames` bytes apart instead of the row pitch the library suggests (polar_hip_row_pitch)")
    ap.add_argument("--force-dist", action="store_true", help="join the RCCL process group even as a world of one (rehearses the N > 1 path on a one-GPU box)")
    ap.add_argument("--fail-rank", type=int, default=-1, help=argparse.SUPPRESS)  # test: this rank exits 3 at start
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------
# rank launcher: `bench.py --gpus N` outside torchrun
# ---------------------------------------------------------------------------------------

def visible_gpus():
    """Number of GPUs this process may use, WITHOUT touching HIP / HSA (the parent must not hold the
    GPU when it starts its children): KFD topology nodes that have SIMDs, narrowed by the
    *_VISIBLE_DEVICES variables. Returns None when the topology cannot be read (then every child
    checks its own LOCAL_RANK against the runtime's count and exits non-zero if it has no GPU)."""
    import glob

    nodes = 0
    files = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not files:
        return None
    for f in files:
        try:
            for line in open(f):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    nodes += 1
        except (OSError, ValueError):
            return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            nodes = min(nodes, len([x for x in v.split(",") if x.strip() != ""]))
    return nodes


def _tail(path, n=25):
    try:
        with open(path, "rb") as f:
            return b"\n".join(f.read().splitlines()[-n:]).decode(errors="replace")
    except OSError:
        return ""


def launch_ranks(args, argv):
    """Start one child per GPU, relay rank 0's line. The parent never initialises HIP and never
    replaces itself: children are ordinary subprocesses. All children are polled: the first one
    that exits non-zero (or the deadline) ends the others, its stderr tail is printed and the
    launcher returns non-zero -- a rank that dies before the rendezvous cannot leave the rest
    waiting in it."""
    import tempfile

    n = args.gpus
    if not args.plumbing_only:
        have = visible_gpus()
        if have is not None and have < n:
            sys.stderr.write("bench.py: --gpus %d asked for but only %d GPU(s) are visible on this node\n" % (n, have))
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    tmp = tempfile.mkdtemp(prefix="bench_ranks_")
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = open(os.path.join(tmp, "rank%d.out" % r), "wb")
        err = open(os.path.join(tmp, "rank%d.err" % r), "wb")
        logs.append((out, err))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out, stderr=err))
    deadline = time.time() + args.rank_timeout
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(rcs) if c not in (None, 0)]
        if bad:
            failed = bad
            break
        if all(c == 0 for c in rcs):
            break
        if time.time() > deadline:
            failed = [(r, "no exit after %.0f s" % args.rank_timeout) for r, c in enumerate(rcs) if c is None]
            break
        time.sleep(0.05)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 5
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    for out, err in logs:
        out.close()
        err.close()
    if failed:
        sys.stderr.write("bench.py: ranks failed (rank, exit code): %s; the other ranks were stopped\n" % failed)
        for r, _ in failed:
            sys.stderr.write("---- rank %d stderr (tail) ----\n%s\n" % (r, _tail(os.path.join(tmp, "rank%d.err" % r))))
        return 1
    sys.stderr.write(_tail(os.path.join(tmp, "rank0.err"), 5) + "\n")
    sys.stdout.write(open(os.path.join(tmp, "rank0.out")).read())
    sys.stdout.flush()
    return 0


# ---------------------------------------------------------------------------------------
# workload
# ---------------------------------------------------------------------------------------

def make_code(polar_amd, M):
    N = 1 << M
    fro = np.zeros(N, np.uint8)
    polar_amd.PolarCodeConst0()(fro, M, N // 2, polar_amd.testbench_probability())
    prog = np.zeros(2 * N + 1, np.uint8)
    n = polar_amd.PolarCompiler()(prog, fro, M)
    return fro, prog[:n].copy()


def generate(torch, polar_amd, sess, fro, M, systematic, frames, seed, first_frame, dev, snr_db=SNR_DB, pitched=False):
    """Resident batch: llr [N][frames] int8 and the decoder-domain (u) truth [K][frames].
    Message bits and noise come from the library's counter-based source (polar_hip_random_message,
    polar_hip_channel): global frame g is the same whatever rank or chunk produces it."""
    N = 1 << M
    K = N // 2
    # pitched: rows polar_hip_row_pitch(frames) bytes apart (a view of a wider allocation), the layout the library
    # suggests for decoder input whose rows would otherwise be a multiple of 4 KiB apart
    llr = polar_amd.pitched_empty(N, frames, dev) if pitched else torch.empty((N, frames), dtype=torch.int8, device=dev)
    u_true = torch.empty((K, frames), dtype=torch.int8, device=dev)
    enc = (polar_amd.PolarSysEnc if systematic else polar_amd.PolarEncoder)(M, sess)
    plain = polar_amd.PolarEncoder(M, sess)
    chunk = max(1, min(frames, (1 << 32) // N))
    stats = torch.zeros(4, dtype=torch.int64, device=dev)
    for a in range(0, frames, chunk):
        c = min(chunk, frames - a)
        msg = torch.empty((K, c), dtype=torch.int8, device=dev)
        sess.random_message(msg, seed, first_frame + a)
        cw = torch.empty((N, c), dtype=torch.int8, device=dev)
        enc(cw, msg, fro)
        q = torch.empty((N, c), dtype=torch.int8, device=dev)
        sess.channel(q, cw, snr_db, seed, first_frame + a)
        sess.count(stats, q, cw, None, None)  # awgn_errors / quantization_erasures of the batch
        llr[:, a:a + c] = q
        if systematic:
            # the u the decoder must return = info rows of the first encoding pass
            # (polar_encoder.hh:40-52) = re-encode + gather of the message itself
            plain.extract_systematic(msg, fro)
        u_true[:, a:a + c] = msg
        del cw, q, msg
    torch.cuda.synchronize(dev)
    return llr, u_true, stats


def systematic_errors(torch, polar_amd, sess, fro, M, out, frames, seed, first_frame, dev):
    """testbench.cc:177-192 after the timed region: re-encode + gather the decoded u, compare with
    the transmitted message (regenerated from the counter-based source), chunk by chunk."""
    N = 1 << M
    K = N // 2
    plain = polar_amd.PolarEncoder(M, sess)
    c4 = torch.zeros(4, dtype=torch.int64, device=dev)
    chunk = max(1, min(frames, (1 << 31) // N))
    for a in range(0, frames, chunk):
        c = min(chunk, frames - a)
        dec = out[:, a:a + c].contiguous()
        plain.extract_systematic(dec, fro)
        msg = torch.empty((K, c), dtype=torch.int8, device=dev)
        sess.random_message(msg, seed, first_frame + a)
        sess.count(c4, None, None, dec, msg)
        del dec, msg
    return c4


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(jobs, seconds):
    """The genuine reference decoder (PolarDecoder<SIMD<int8_t,32>>) on the host cores. The GPU
    box gives a 1-GPU job a share of 16 CPUs (OMP_NUM_THREADS there), so 16 threads are used
    even though os.cpu_count() reports the whole machine. jobs: [(llr_sample, prog, K)], one per
    code of the workload (one for configs A / B / D); every thread decodes bundles of every code,
    the time budget split evenly between the codes."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or 16
    threads = max(1, min(share, os.cpu_count() or 1))
    if os.path.exists(oracle_lib.REF_SO):
        ref = oracle_lib.Ref()
        W = ref.width
        kind = "reference"

        def timed(bundle, prog, reps):
            return ref.decode_time(bundle, prog, reps)
    else:
        orc = oracle_lib.Oracle()
        W = 32
        kind = "port"

        def timed(bundle, prog, reps):
            return orc.decode_time(bundle, prog, reps)
    plan = []  # per code: (bundles per thread, prog, K, reps)
    for llr_sample, prog, K in jobs:
        have = llr_sample.shape[1] // W
        bundles = [np.ascontiguousarray(llr_sample[:, (t % have) * W:(t % have + 1) * W]) for t in range(threads)]
        per_call = timed(bundles[0], prog, 3) / 3
        plan.append((bundles, prog, K, max(1, int(seconds / len(jobs) / per_call))))
    out = [0.0] * threads

    def work(t):
        out[t] = sum(timed(bundles[t], prog, reps) for bundles, prog, _K, reps in plan)

    th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    t0 = time.time()
    for x in th:
        x.start()
    for x in th:
        x.join()
    wall = time.time() - t0
    bits = threads * sum(reps * W * K for _b, _p, K, reps in plan)
    calls = sum(reps for _b, _p, _K, reps in plan)
    return {
        "value": bits / max(out) / 1e9, "unit": "Gbit/s", "cores": threads, "kind": kind,
        "host_cpus_visible": os.cpu_count(), "cpu_model": cpu_model(),
        "sample": "%d threads x %d calls x %d frames over %d code(s) of the workload, the same LLR bundle per code and thread, decode only (wall %.1f s, %.1f us per call); threads = the job's CPU share, not the machine's %s CPUs" % (
            threads, calls, W, len(jobs), wall, 1e6 * max(out) / calls, os.cpu_count()),
    }


def _file_sha16(path):
    import hashlib

    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def library_sha16():
    """Identity of the kernel code the counters were (or are) taken on: the decode-kernel sources."""
    import hashlib

    h = hashlib.sha256()
    for f in ("sc_core.hpp", "decode_kernel.hpp", "variants.hip", "program.hpp", "decode_launch.hpp", "handle.hpp"):
        h.update(open(os.path.join(ROOT, "polar_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def load_profile_counters(frames, args):
    """HBM bytes and VALU / LDS utilisation of the decode kernel from the committed rocprofv3 PMC
    passes of this same command (tools/prof_bench.sh -> profiles/hbm_traffic*.json). These are NOT
    measured in this run: the record says which file they come from and on which kernel sources
    they were taken (`kernel_sources_sha16`); when the sources have changed since, the figures are
    dropped from the line and `stale` is set instead."""
    for name in ("hbm_traffic.json", "hbm_traffic_%s.json" % args.config):
        tpath = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(tpath):
            continue
        t = json.load(open(tpath))
        if t.get("frames") == frames and t.get("fpw") == args.fpw and t.get("lds_level") == args.lds_level and t.get("config", "B") == args.config and "bytes_per_launch" in t:
            t["_file"] = "profiles/" + name
            t["_stale"] = t.get("kernel_sources_sha16") != library_sha16()
            return t
    return None


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.fail_rank == rank:
        sys.stderr.write("bench.py: rank %d fails on purpose (--fail-rank)\n" % rank)
        sys.exit(3)
    import torch

    if args.plumbing_only:
        return plumbing_only(torch, args, rank, world)
    dist = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: 0 GPU(s) are visible to rank %d and the HIP path has no CPU fallback" % rank)
    if torch.cuda.device_count() <= local:
        raise SystemExit("bench.py: rank %d needs GPU %d but only %d GPU(s) are visible" % (rank, local, torch.cuda.device_count()))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1 or args.force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:  # --force-dist on its own: a world of one needs the rendezvous variables too
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        import datetime

        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=args.init_timeout))
    import polar_amd

    M, systematic, default_frames = CONFIGS[args.config]
    sess = polar_amd.Session(local)
    if args.fpw:
        sess.configure(args.fpw, args.lds_level, 0)
    if args.config == "E":
        run_config_e(torch, polar_amd, sess, args, rank, world, dev, dist)
        if dist is not None:
            dist.destroy_process_group()
        return
    N = 1 << M
    K = N // 2
    frames = args.frames or default_frames
    fro, prog = make_code(polar_amd, M)
    seed, first = 42, rank * frames
    llr, u_true, chan = generate(torch, polar_amd, sess, fro, M, systematic, frames, seed, first, dev, pitched=not args.contiguous_llr)
    out = torch.empty((K, frames), dtype=torch.int8, device=dev)
    counters = torch.zeros(4, dtype=torch.int64, device=dev)
    dec = polar_amd.PolarDecoder(sess)
    plain = polar_amd.PolarEncoder(M, sess)
    kernel_ms = []

    def step():
        dec(out, llr, prog)
        kernel_ms.append(sess.last_decode_ms())
        if args.include_extract:
            plain.extract_systematic(out, fro)

    def tally():
        # testbench.cc:189-192 once per batch, then the one collective of the job
        counters.zero_()
        sess.count(counters, None, None, out, msg_true if args.include_extract else u_true)
        if dist is not None:
            dist.all_reduce(counters)

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    msg_true = None
    if args.include_extract:
        msg_true = torch.empty((K, frames), dtype=torch.int8, device=dev)
        sess.random_message(msg_true, seed, first)
    # Part of the set-up, before the warm-up steps: the library looks at a few placements of its scratch
    # arenas, one call each, and keeps the fastest (decode_launch.hpp, placement_step: where the driver puts
    # 20 GiB of arenas decides up to 5 % of the kernel time), then runs one call as relay launches and keeps
    # them if they are faster. SETTLE_CALLS calls see both through.
    for _ in range(SETTLE_CALLS):
        dec(out, llr, prog)
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
    if systematic and not args.include_extract:
        sysc = systematic_errors(torch, polar_amd, sess, fro, M, out, frames, seed, first, dev)
    else:
        sysc = counters.clone()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.all_reduce(chan)
        if systematic and not args.include_extract:
            dist.all_reduce(sysc)
    c = counters.cpu().tolist()
    sc = sysc.cpu().tolist()
    ch = chan.cpu().tolist()
    total_frames = frames * world
    value = total_frames * K * args.steps / elapsed / 1e9
    if rank == 0:
        k_ms = float(np.mean(kernel_ms))
        shape = sess.last_decode_shape()
        alg = frames * 1.5 * N
        achieved = alg / (k_ms * 1e-3) / 1e9
        prof = load_profile_counters(frames, args)
        if prof and prof["_stale"]:
            stale, prof = prof, None
        else:
            stale = None
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": prof.get("bytes_per_launch") if prof else None,
                "kernel": ("sc_decode_team_kernel<%d, %d, %d>" % (shape[0] // 4, shape[1], shape[2])) if shape[2] > 1 else "sc_decode_kernel<%d, %d>" % (shape[0] // 4, shape[1]),
                "kernel_ms": k_ms,
                "algorithmic_bytes_per_launch": alg}
        launches = sess.last_decode_launches()
        if launches > 1:
            # one decode of the batch = `launches` relay launches of half a grid each on two streams (decode_launch.hpp);
            # they overlap, so the unit of this block is the decode: kernel_ms runs from the start of the first
            # launch to the end of the last (HIP events on the launch stream, around the fork and the join)
            roof["launches_per_decode"] = launches
            roof["note"] = "launch = one decode of the batch, sent as %d overlapping relay launches; kernel_ms = first start to last end" % launches
        if prof:
            # counters of the same command under rocprofv3 (profiles/): real HBM bytes per launch
            # over the live kernel time, and the VALU / LDS figures north_star asks for
            roof["traffic_frac"] = prof["bytes_per_launch"] / (k_ms * 1e-3) / (HBM_PEAK_GBS * 1e9)
            for key in ("valu_util", "lds_bw_util", "lds_bank_conflict_frac"):
                if key in prof:
                    roof[key] = prof[key]
            roof["counters_from"] = {"file": prof["_file"], "kernel_sources_sha16": prof.get("kernel_sources_sha16"),
                                     "note": "rocprofv3 --pmc passes of this command (tools/prof_bench.sh), committed; not measured in this run"}
        elif stale:
            roof["counters_from"] = {"file": stale["_file"], "stale": True, "kernel_sources_sha16": stale.get("kernel_sources_sha16"),
                                     "current_sources_sha16": library_sha16(),
                                     "note": "the committed counters were taken on other kernel sources: traffic / VALU / LDS figures left out"}
        what = "SC decode of the batch (what testbench.cc:170-174 times)"
        if args.include_extract:
            what += " + fused re-encode/gather (testbench.cc:177-183)"
        line = {
            "metric": "decoded info Gbit/s (node) at N=2^%d R=1/2 int8 SC" % M, "value": value, "unit": "Gbit/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int8", "data": "synthetic",
            "config": {
                "workload": "config %s: N=2^%d K=%d %s Const0 code, %d frames per GPU resident in HBM, BPSK/AWGN Es/N0=%.1f dB, int8 LLRs; step = %s; the timed region ends with one error count%s" % (
                    args.config, M, K, "systematic" if systematic else "non-systematic", frames, SNR_DB, what, " + RCCL all-reduce of the counters" if world > 1 else ""),
                "frames_per_gpu": frames, "parallelism": "frames sharded over %d GPU(s)" % world,
                "frames_per_wave": shape[0], "lds_level": shape[1], "team": shape[2], "shape_chosen_by": "caller" if args.fpw else "library default", "setup_decode_calls": SETTLE_CALLS,
                "llr_row_pitch_bytes": int(llr.stride(0)),
            },
            "roofline": roof,
            "frac_of_hbm_roofline_whole_job": value * 1e9 * 3 / (HBM_PEAK_GBS * 1e9 * world),
            # ber: the reference's figure (systematic: after re-encode + gather, testbench.cc:177-197);
            # ber_u_domain: decoder output against the encoder's u, what the timed tally counts
            "ber": sc[2] / float(total_frames * K), "uncorrected_errors": sc[2], "ambiguity_erasures": sc[3],
            "ber_u_domain": c[2] / float(total_frames * K),
            "awgn_errors": ch[0], "quantization_erasures": ch[1],
        }
        if world == 1 and args.cpu_seconds > 0:
            sample = llr[:, :16 * 32].cpu().numpy() if M <= 16 else llr[:, :32].cpu().numpy()
            line["cpu_baseline"] = cpu_baseline([(sample, prog, K)], args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def run_config_e(torch, polar_amd, sess, args, rank, world, dev, dist):
    """BASELINE configs[4]: one launch set that mixes N = 2^10 .. 2^16, every (N, channel SNR) batch with
    the frozen set PolarCodeConst0 designs for that very SNR (testbench.cc:74-88 evaluated per point,
    through the handle's (level, K, p) cache: construction, compilation and upload happen once per
    key). Every batch holds the same number of LLR bytes, so small codes bring many frames and the
    largest code few (the small-batch regime of the decoder). One step = the decode of all batches,
    back to back; `value` = information bits of all of them / wall time."""
    from polar_amd.sweep import design_probability

    batch_bytes = args.frames or CONFIGS["E"][2]
    seed = 42
    jobs = []
    t_code = time.perf_counter()
    for M in E_LEVELS:
        N = 1 << M
        for si, snr in enumerate(E_SNRS):
            fro, prog, _amb = sess.code(M, N // 2, design_probability(snr))
            jobs.append({"M": M, "snr": snr, "fro": fro, "prog": prog, "frames": max(32, batch_bytes // N), "si": si})
    t_code = time.perf_counter() - t_code
    chan = torch.zeros(4, dtype=torch.int64, device=dev)
    for j in jobs:
        # frames of different (N, SNR) batches never share a counter-based stream: the frame index space is split per job
        first = (jobs.index(j) * world + rank) * (1 << 24)
        j["llr"], j["u"], st = generate(torch, polar_amd, sess, j["fro"], j["M"], 1, j["frames"], seed, first, dev, j["snr"], pitched=not args.contiguous_llr and args.streams == 0)
        j["out"] = torch.empty(((1 << j["M"]) // 2, j["frames"]), dtype=torch.int8, device=dev)
        chan += st
    counters = torch.zeros(4, dtype=torch.int64, device=dev)
    kernel_ms, shapes = [], {}
    # The launch set runs on `lanes` handles, each with a stream of its own and an equal share of the
    # resident wavefronts (polar_hip_configure max_waves: the shape is picked for that share), batch i
    # on lane i % lanes: batches of the large codes are few frames each and bound by the serial depth
    # of the program, so several of them side by side fill the device that one alone leaves idle.
    use_set = args.streams <= 0
    lanes = 4 if use_set else args.streams
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    sessions = [sess] + ([] if use_set else [polar_amd.Session(dev.index) for _ in range(lanes - 1)])
    stride = int(os.environ.get("BENCH_STREAM_STRIDE", "1"))  # experiments: which of torch's pooled streams the lanes take
    streams = [] if use_set else [torch.cuda.Stream(device=dev) for _ in range(lanes * stride)][::stride]
    decs = [polar_amd.PolarDecoder(s) for s in sessions]
    if lanes > 1 and not args.fpw:
        for s in sessions:
            s.configure(0, 0, args.lane_waves or 8 * cus // lanes)
    # every batch decoded once alone on a share-sized handle: the shape it gets, and (lanes managed here)
    # its solo time for the placement: longest first onto the lane with the least work so far
    for j in jobs:
        decs[0](j["out"], j["llr"], j["prog"])
        j["solo_ms"] = sessions[0].last_decode_ms()
        j["shape"] = sessions[0].last_decode_shape()
    if use_set and not args.fpw:
        sess.configure(0, 0, 0)
    load = [0.0] * lanes
    order = sorted(range(len(jobs)), key=lambda i: -jobs[i]["solo_ms"])
    if not use_set:
        for i in order:
            k = load.index(min(load))
            jobs[i]["lane"] = k
            load[k] += jobs[i]["solo_ms"]
            sessions[k].load_program(jobs[i]["prog"])
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    set_jobs = [(j["out"], j["llr"], j["prog"]) for j in jobs]

    def step():
        torch.cuda.synchronize(dev)
        if use_set:  # one library call: polar_hip_decode_set places and launches the batches on its lanes
            cur = torch.cuda.current_stream(dev)
            ev0.record(cur)
            decs[0].decode_set(set_jobs)
            ev1.record(cur)
        else:
            ev0.record(streams[0])
            for st in streams[1:]:
                st.wait_event(ev0)
            for i in order:
                j = jobs[i]
                decs[j["lane"]].launch(j["out"], j["llr"], j["prog"], streams[j["lane"]].cuda_stream)
            for st in streams[1:]:
                streams[0].wait_stream(st)
            ev1.record(streams[0])
        torch.cuda.synchronize(dev)
        kernel_ms.append(ev0.elapsed_time(ev1))  # GPU time of the whole launch set (the launches overlap)
        for j in jobs:
            shapes.setdefault(j["M"], j["shape"])

    def tally():
        counters.zero_()
        for j in jobs:
            sess.count(counters, None, None, j["out"], j["u"])
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
    info_bits = sum(j["frames"] * (1 << j["M"]) // 2 for j in jobs)
    value = info_bits * world * args.steps / elapsed / 1e9
    if rank != 0:
        return
    k_ms = float(np.mean(kernel_ms))
    alg = sum(j["frames"] * 1.5 * (1 << j["M"]) for j in jobs)
    achieved = alg / (k_ms * 1e-3) / 1e9
    prof = load_profile_counters(batch_bytes, args)
    stale = None
    if prof and prof["_stale"]:
        stale, prof = prof, None
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": prof.get("bytes_per_launch") if prof else None,
            "kernel": "sc_decode kernels, one launch per batch; shape (frames per wavefront, LDS level, team) by N: " + ", ".join(
                "2^%d: %s" % (M, tuple(shapes[M])) for M in sorted(shapes)),
            "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg,
            "note": "one 'launch' here = the %d decode launches of a step, spread over %d HIP streams (%s); kernel_ms = GPU time of the whole set (events around it)" % (len(jobs), lanes, "polar_hip_decode_set" if use_set else "handles managed by bench.py")}
    if prof:
        roof["traffic_frac"] = prof["bytes_per_launch"] / (k_ms * 1e-3) / (HBM_PEAK_GBS * 1e9)
        for key in ("valu_util", "lds_bw_util", "lds_bank_conflict_frac"):
            if key in prof:
                roof[key] = prof[key]
        roof["counters_from"] = {"file": prof["_file"], "kernel_sources_sha16": prof.get("kernel_sources_sha16"),
                                 "note": "rocprofv3 --pmc passes of this command (tools/prof_bench.sh), committed; not measured in this run"}
    elif stale:
        roof["counters_from"] = {"file": stale["_file"], "stale": True, "note": "taken on other kernel sources: figures left out"}
    line = {
        "metric": "decoded info Gbit/s (node), mixed N=2^10..2^16 R=1/2 int8 SC with per-SNR frozen sets", "value": value, "unit": "Gbit/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int8", "data": "synthetic",
        "config": {
            "workload": "config E: %d resident batches per GPU = N in 2^10..2^16 x channel Es/N0 in %s dB, each with the systematic Const0 code designed for its own SNR (per-SNR frozen set and program from the handle's cache), %d MiB of int8 LLRs per batch (%d .. %d frames); step = SC decode of every batch, the launch set spread over %d handles / HIP streams; the timed region ends with one error count%s" % (
                len(jobs), list(E_SNRS), batch_bytes >> 20, jobs[-1]["frames"], jobs[0]["frames"], lanes, " + RCCL all-reduce of the counters" if world > 1 else ""),
            "batches_per_gpu": len(jobs), "batch_bytes": batch_bytes, "parallelism": "frames sharded over %d GPU(s)" % world,
            "shape_chosen_by": "caller" if args.fpw else "library default",
            "code_construction_s": t_code,
        },
        "roofline": roof,
        "frac_of_hbm_roofline_whole_job": value * 1e9 * 3 / (HBM_PEAK_GBS * 1e9 * world),
        "ber_u_domain": c[2] / float(info_bits * world), "uncorrected_errors_u_domain": c[2], "ambiguity_erasures": c[3],
        "awgn_errors": ch[0], "quantization_erasures": ch[1],
    }
    if world == 1 and args.cpu_seconds > 0:
        sample = [(j["llr"][:, :64].cpu().numpy(), j["prog"], (1 << j["M"]) // 2) for j in jobs if j["si"] == len(E_SNRS) // 2]
        line["cpu_baseline"] = cpu_baseline(sample, args.cpu_seconds)
    print(json.dumps(line), flush=True)


def plumbing_only(torch, args, rank, world):
    """The rank plumbing of main() with CPU tensors over gloo (tests/test_multi_rank.py): process
    group from the environment, barrier-bracketed timed region, counter all-reduce, max-over-ranks
    time, one line from rank 0. No decode happens and `value` is 0."""
    import torch.distributed as dist

    from polar_amd.shard import shard_range

    if world > 1:
        import datetime

        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=args.init_timeout))
    frames = args.frames or 1000
    a, b = shard_range(frames * world, rank, world)
    counters = torch.tensor([rank + 1, b - a, 0, 0], dtype=torch.int64)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    if world > 1:
        dist.all_reduce(counters)
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "plumbing only", "value": 0.0, "n_gpus": world, "counters": counters.tolist(),
                          "ms_per_step": 1e3 * float(t.item()), "plumbing_only": True}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
