"""Multi-GPU sharding of independent OfflineAudioContext batches.

Contexts never exchange data, so the N-GPU path is: contiguous instance ranges per device, one batch per (device,
sub-batch), no data-path collective.  Two layers:

* ``shard_range`` / ``timed_steps`` — the partition and the measurement protocol of bench.py (barrier + MAX over ranks
  of the elapsed time; any torch.distributed backend: nccl = RCCL on the GPUs, gloo in the CPU tests).
* ``render_sharded`` — the N-device render component: takes the HOST buffers of all contexts, splits them into
  contiguous ranges per device and sub-batches per device, and pipelines  upload(k+1) || render(k) || download(k-1)
  on every device (one host thread per sub-batch — ctypes releases the GIL —, every batch has its own HIP stream and its
  bulk transfers run on it; one upload and one download per device at a time, in index order, because the link is full
  duplex but two uploads only share it).  This is what a caller holding 4096 contexts' AudioBuffers on the host runs
  instead of 4096 x start_rendering_sync; under torch.distributed every rank calls it with its own device and shard.
"""
from __future__ import annotations

import ctypes as C
import threading
import time
from typing import Callable, Optional, Sequence, Tuple

import numpy as np


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous range [lo, hi) of instances owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def timed_steps(step: Callable[[], None], sync: Callable[[], None], steps: int, warmup: int, dist=None,
                device_tensor: Optional[Callable[[float], object]] = None,
                after_warmup: Optional[Callable[[], None]] = None) -> float:
    """bench.py's protocol: W untimed steps, barrier+sync, exactly K timed steps, sync+barrier, MAX over ranks.
    `after_warmup` runs between the warm-up and the timed region, device idle (bench.py resets the per-kernel event
    totals there, so that kernel averages cover the timed steps only)."""
    for _ in range(warmup):
        step()
    sync()
    if after_warmup is not None:
        after_warmup()
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


class _Turn:
    """Sub-batches of one device take the link in index order (one direction each)."""

    def __init__(self):
        self.cv, self.next = threading.Condition(), 0

    def wait(self, k: int):
        with self.cv:
            self.cv.wait_for(lambda: self.next == k)

    def done(self):
        with self.cv:
            self.next += 1
            self.cv.notify_all()


def _addr(a) -> int:
    """Host address of a numpy array or a (pinned) torch tensor."""
    return a.data_ptr() if hasattr(a, "data_ptr") else a.ctypes.data


def plan_shards(n_total: int, devices: Sequence[int], sub_batches: int):
    """[(slot, device, k, lo, hi)]: contiguous instance ranges per entry of `devices` (shard_range; slot = its position — a
    device may be listed twice), each split into at most `sub_batches` contiguous sub-batches (k = its index in that
    slot); empty ranges are dropped."""
    out = []
    for di, dev in enumerate(devices):
        lo, hi = shard_range(n_total, di, len(devices))
        parts = max(1, min(sub_batches, hi - lo))
        k = 0
        for p in range(parts):
            a, b = shard_range(hi - lo, p, parts)
            if b > a:
                out.append((di, dev, k, lo + a, lo + b))
                k += 1
    return out


def render_sharded(build: Callable, host_in, host_out, devices: Sequence[int] = (0,), sub_batches: int = 8,
                   sample_rate: float = 48000.0, pcm16: bool = False, pull: Optional[Callable] = None):
    """Render N contexts whose source AudioBuffers live on the host, on one or several GPUs.

    build(n_instances, device) -> (ctx, src): builds the (identical) graph for a sub-batch of `n_instances` contexts on
        `device`; `src` is the AudioBufferSourceNode that receives the contexts' buffers.  The ctx must not be rendered.
    host_in:  [N, channels, frames] float32 — or, with pcm16=True, [N, frames, channels] int16 (decoded WAV data: half
        the upload, converted on the device) — numpy array or pinned torch tensor.
    host_out: [N, n_out, length] float32, filled with every context's rendered AudioBuffer.
    pull(ctx, lo, hi): optional control-side work per sub-batch after its render (e.g. the batched analyser pull).
    Returns {"seconds": wall time, "shards": [(device, lo, hi), ...]}.  Raises the first sub-batch error."""
    n_total = int(host_in.shape[0])
    assert int(host_out.shape[0]) == n_total
    row_in = int(np.prod(host_in.shape[1:])) * (2 if pcm16 else 4)
    row_out = int(np.prod(host_out.shape[1:])) * 4
    frames = int(host_in.shape[1] if pcm16 else host_in.shape[2])
    n_ch = int(host_in.shape[2] if pcm16 else host_in.shape[1])
    shards = plan_shards(n_total, list(devices), sub_batches)
    up = [_Turn() for _ in devices]
    down = [_Turn() for _ in devices]
    errors = []
    FP, I16 = C.POINTER(C.c_float), C.POINTER(C.c_int16)
    base_in, base_out = _addr(host_in), _addr(host_out)

    def run(slot, dev, k, lo, hi):
        ctx = None
        took_up = took_down = False
        try:
            ctx, src = build(hi - lo, dev)
            ctx.prepare()
            b = ctx._b
            up[slot].wait(k)
            took_up = True
            try:
                if pcm16:
                    b.check(b.source_set_buffer_pcm16_batch(ctx._handle, src.id, C.cast(base_in + lo * row_in, I16), n_ch, frames,
                                                           sample_rate))
                else:
                    b.check(b.source_set_buffer_batch(ctx._handle, src.id, C.cast(base_in + lo * row_in, FP), n_ch, frames,
                                                     sample_rate))
            finally:
                up[slot].done()
            b.check(b.render(ctx._handle))
            b.check(b.sync(ctx._handle))
            if pull is not None:
                pull(ctx, lo, hi)
            down[slot].wait(k)
            took_down = True
            try:
                b.check(b.download_all(ctx._handle, C.cast(base_out + lo * row_out, FP)))
            finally:
                down[slot].done()
        except Exception as e:  # noqa: BLE001 — reported to the caller below
            errors.append(e)
            if not took_up:
                up[slot].wait(k)
                up[slot].done()
            if not took_down:
                down[slot].wait(k)
                down[slot].done()
        finally:
            if ctx is not None:
                ctx.close()

    t0 = time.perf_counter()
    threads = [threading.Thread(target=run, args=s) for s in shards]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return {"seconds": time.perf_counter() - t0, "shards": [(d, lo, hi) for _, d, _, lo, hi in shards]}
