#!/usr/bin/env python3
"""BER curves in the reference's output format (SNR BER Mbit/s Eb/N0, testbench.cc:218) from
the device simulation: systematic / non-systematic x code length, the curves of README.md:24.

    python tools/ber_sweep.py --M 14 16 --max-frames 262144 [--non-systematic] [--per-snr-design]
    python -m torch.distributed.run --nproc-per-node 8 tools/ber_sweep.py ...     (frames sharded, counters all-reduced)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, nargs="+", default=[14])
ap.add_argument("--max-frames", type=int, default=1 << 16)
ap.add_argument("--chunk", type=int, default=1 << 14)
ap.add_argument("--seed", type=int, default=42)
ap.add_argument("--non-systematic", action="store_true")
ap.add_argument("--both", action="store_true", help="systematic and non-systematic curves")
ap.add_argument("--per-snr-design", action="store_true")
a = ap.parse_args()
import torch

rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
dist = None
torch.cuda.set_device(local)
if world > 1:
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
import polar_amd
from polar_amd.sweep import ber_sweep

sess = polar_amd.Session(local)
for M in a.M:
    for systematic in ((True, False) if a.both else ((not a.non_systematic),)):
        ber_sweep(sess, M, systematic=systematic, seed=a.seed, chunk=min(a.chunk, a.max_frames), max_frames=a.max_frames,
                  per_snr_design=a.per_snr_design, dist=dist, rank=rank, world=world)
if dist is not None:
    dist.destroy_process_group()
