"""Worker for tests/test_multi_rank.py: one process per rank (gloo), each renders its shard with the CPU
oracle standing in for the device library (the sharding logic is backend independent)."""
import ctypes
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import web_audio_api_rs_amd as waa  # noqa: E402
from graphs import c2, white_noise  # noqa: E402
from web_audio_api_rs_amd.sharding import render_sharded, shard_range, timed_steps  # noqa: E402


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    orc = waa.bind(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")), "orc_")
    n_total, frames = 7, 128 * 20 + 3
    lo, hi = shard_range(n_total, rank, world)
    noise = white_noise(hi - lo, 2, frames, first=lo)
    out = {}

    def step():
        ctx, _ = c2(orc, noise)
        out["data"] = ctx.start_rendering_sync().data
        ctx.close()

    elapsed = timed_steps(step, lambda: None, steps=2, warmup=1, dist=dist)
    sums = torch.zeros(n_total, dtype=torch.float64)
    sums[lo:hi] = torch.from_numpy(out["data"].astype(np.float64).sum(axis=(1, 2)))
    dist.all_reduce(sums)  # checksum gather (test only; the data path itself has no collective)
    # the same shard through the N-device render component (host buffers in, host buffers out, two pipelined sub-batches)
    out2 = np.zeros((hi - lo, 2, frames), np.float32)

    def build(n, device):
        ctx, nodes = c2(orc, np.zeros((n, 2, frames), np.float32))  # (the buffers arrive through render_sharded)
        return ctx, nodes["src"]

    render_sharded(build, noise, out2, devices=[-1], sub_batches=2)
    sums2 = torch.zeros(n_total, dtype=torch.float64)
    sums2[lo:hi] = torch.from_numpy(out2.astype(np.float64).sum(axis=(1, 2)))
    dist.all_reduce(sums2)
    if rank == 0:
        print(json.dumps({"elapsed": elapsed, "sums": sums.tolist(), "sums_sharded": sums2.tolist(), "ranges": [shard_range(n_total, r, world) for r in range(world)]}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
