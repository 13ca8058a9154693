"""Frame sharding + counter all-reduce with two gloo ranks on the CPU.

The decode itself needs the GPU; what is checked here is the N>1 host logic bench.py relies
on: contiguous frame shards cover the batch exactly once for any world size, and the int64[4]
counters summed by all_reduce equal the counters of the whole batch. The per-shard counters
come from the oracle (this is a test; the product path has no CPU fallback)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from polar_amd.shard import shard_range


def test_shard_ranges_partition():
    for frames in (0, 1, 7, 64, 1000, 1 << 20):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(frames, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == frames
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 4 and all(a % 4 == 0 for a, _ in spans)


def _worker(rank, world, port, noisy, orig, decoded, message, expected, q):
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from oracle_lib import Oracle

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = noisy.shape[1]
    a, b = shard_range(frames, rank, world)
    c = Oracle().count(np.ascontiguousarray(noisy[:, a:b]), np.ascontiguousarray(orig[:, a:b]),
                       np.ascontiguousarray(decoded[:, a:b]), np.ascontiguousarray(message[:, a:b]))
    t = torch.from_numpy(c.copy())
    dist.all_reduce(t)
    # the counter-based sources (include/polar_hip.h) give each rank exactly its slice of the batch
    o = Oracle()
    msg = o.random_message(8, b - a, 42, first_frame=a)
    llr = o.channel(msg, -1.0, 42, first_frame=a)
    whole_msg = o.random_message(8, frames, 42)
    same = (msg == whole_msg[:, a:b]).all() and (llr == o.channel(whole_msg, -1.0, 42)[:, a:b]).all()
    q.put((rank, t.tolist() == expected and bool(same)))
    dist.destroy_process_group()


def test_two_rank_counter_allreduce(oracle):
    rng = np.random.default_rng(0)
    frames = 50
    noisy = rng.integers(-3, 4, (64, frames)).astype(np.int8)
    orig = (1 - 2 * rng.integers(0, 2, (64, frames))).astype(np.int8)
    decoded = rng.integers(-1, 2, (32, frames)).astype(np.int8)
    message = (1 - 2 * rng.integers(0, 2, (32, frames))).astype(np.int8)
    expected = oracle.count(noisy, orig, decoded, message).tolist()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, noisy, orig, decoded, message, expected, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, True), (1, True)]


def test_bench_launcher_starts_the_ranks():
    """`bench.py --gpus N` outside torchrun must start N ranks itself (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* plumbing), run the barrier-bracketed region and the counter all-reduce,
    and print one line with n_gpus == N. Exercised here over gloo on the CPU."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--plumbing-only", "--frames", "1001"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2
    assert d["counters"][0] == 1 + 2 and d["counters"][1] == 2002  # both ranks' shards, summed
    # a world size that does not match --gpus is refused instead of printing n_gpus: 1
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--plumbing-only"],
                       capture_output=True, text=True, timeout=120, env=env2)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_bench_launcher_refuses_missing_gpus():
    """Fewer visible devices than --gpus: non-zero exit with a clear message (this container has none)."""
    import subprocess
    import sys

    import torch

    if torch.cuda.device_count() >= 2:
        import pytest

        pytest.skip("needs a machine with fewer than 2 GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "GPU(s) are visible" in r.stderr


def test_bench_launcher_fails_fast_when_a_rank_dies():
    """A rank that exits before the rendezvous must not leave the others (and the launcher) waiting
    in it: the launcher stops the rest, prints the failing rank's stderr and returns non-zero."""
    import subprocess
    import sys
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--plumbing-only", "--fail-rank", "1",
                        "--init-timeout", "600"], capture_output=True, text=True, timeout=120, env=env)
    took = time.time() - t0
    assert r.returncode != 0
    assert "(1, 3)" in r.stderr and "fails on purpose" in r.stderr, r.stderr
    assert took < 60, took  # far below the 600 s the surviving rank would wait in the rendezvous
    # a launcher deadline ends ranks that never finish
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--plumbing-only", "--fail-rank", "1",
                        "--rank-timeout", "0"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and time.time() - t0 < 60


class _FakeSession:
    """Stands in for polar_amd.Session in the sweep harness tests: `simulate` reports one
    uncorrected error per frame below -1.0 dB (per 64 frames between -1.0 and -0.5 dB), none above."""

    def __init__(self):
        self.calls = []

    def code(self, M, K, p):
        return np.zeros(1 << M, np.uint8), np.array([M, 3, 255], np.uint8), False

    def simulate(self, counters, program, frozen, level, systematic, snr_db, seed, first_frame, frames):
        self.calls.append((round(snr_db, 6), first_frame, frames))
        err = frames if snr_db < -1.0 - 1e-9 else (frames // 64 if snr_db < -0.5 - 1e-9 else 0)
        counters += torch.tensor([0, 0, err, 0, frames, frames * (1 << level) // 2], dtype=torch.int64)

    def last_decode_ms(self):
        return 1.0


def test_sweep_harness_follows_the_reference_loop():
    """testbench.cc:105-110, 125, 197-201: points from floor(design - 3) in 0.1 dB steps, the 1000-error
    budget per point, and the stop after the run of clean points that follows the last bad one."""
    import io

    from polar_amd.sweep import ber_sweep, design_snr, snr_points

    pts = list(snr_points(design_snr()))
    assert pts[0] == -5.0 and abs(pts[1] + 4.9) < 1e-12 and pts[-1] < 4.0 + 1e-9
    s = _FakeSession()
    out = io.StringIO()
    rows = ber_sweep(s, 10, chunk=256, max_frames=4096, out=out, device=torch.device("cpu"))
    snrs = [r[0] for r in rows]
    assert snrs[0] == -5.0
    bad = [r for r in rows if r[5][2] > 0]
    assert abs(bad[-1][0] + 0.6) < 1e-9              # last erroneous point
    assert len(rows) == len(bad) + 3                  # then three clean points, then stop (count <= 3)
    assert rows[0][4] == 1024                         # 1000-error budget: four chunks of 256 at one error per frame
    assert rows[len(bad) - 1][4] == 4096              # few errors: the whole frame budget
    assert all(abs(r[1] - r[5][2] / r[5][5]) < 1e-15 for r in rows)
    text = out.getvalue().splitlines()
    assert text[1] == "# SNR BER Mbit/s Eb/N0" and len(text[2].split()) == 4 and text[-1].startswith("# QEF at: -0.5")


def _sweep_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import io

    from polar_amd.sweep import ber_sweep

    s = _FakeSession()
    rows = ber_sweep(s, 10, chunk=256, max_frames=4096, out=io.StringIO(), device=torch.device("cpu"), dist=dist, rank=rank, world=world)
    q.put((rank, [(r[4], r[5][2]) for r in rows], s.calls[:2]))
    dist.destroy_process_group()


def test_sweep_harness_two_ranks():
    """Frames of a chunk are sharded over the ranks, the counters all-reduced between chunks and the
    error budget evaluated on the reduced value: both ranks see the single-rank totals."""
    import io

    from polar_amd.sweep import ber_sweep

    single = ber_sweep(_FakeSession(), 10, chunk=256, max_frames=4096, out=io.StringIO(), device=torch.device("cpu"))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sweep_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    want = [(r[4], r[5][2]) for r in single]
    assert results[0][1] == want and results[1][1] == want
    # rank 0 took frames [0, 128) and rank 1 [128, 256) of the first chunk of the first point
    assert results[0][2][0][1:] == (0, 128) and results[1][2][0][1:] == (128, 128)
