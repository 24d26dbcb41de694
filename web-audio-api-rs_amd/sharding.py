"""Multi-GPU sharding of independent OfflineAudioContext batches.

Contexts never exchange data, so the N-GPU path is: contiguous instance ranges per rank, one batch per
rank/GPU, no data-path collective.  The only communication is the measurement protocol of bench.py
(barrier + MAX over ranks of the elapsed time).  Works with any torch.distributed backend (nccl = RCCL
on the GPUs, gloo in the CPU tests)."""
from __future__ import annotations

import time
from typing import Callable, Optional, Tuple


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous range [lo, hi) of instances owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def timed_steps(step: Callable[[], None], sync: Callable[[], None], steps: int, warmup: int, dist=None,
                device_tensor: Optional[Callable[[float], object]] = None) -> float:
    """bench.py's protocol: W untimed steps, barrier+sync, exactly K timed steps, sync+barrier, MAX over ranks."""
    for _ in range(warmup):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = device_tensor(elapsed) if device_tensor else torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed
