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
