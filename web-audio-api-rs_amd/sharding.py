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


def _host_block(a, name, dtype, shape_tail=None):
    """(address, shape) of a C-contiguous host array of `dtype` — numpy array or (pinned) torch tensor; anything else is an
    error here rather than an out-of-bounds transfer inside the library."""
    if hasattr(a, "data_ptr"):  # torch tensor
        import torch
        want = {np.float32: torch.float32, np.int16: torch.int16}[dtype]
        if a.dtype != want or not a.is_contiguous() or a.device.type != "cpu":
            raise ValueError(f"{name}: expected a contiguous CPU tensor of {want}, got {a.dtype} on {a.device}")
    else:
        if not isinstance(a, np.ndarray) or a.dtype != dtype or not a.flags["C_CONTIGUOUS"]:
            raise ValueError(f"{name}: expected a C-contiguous {np.dtype(dtype).name} array")
    shape = tuple(int(x) for x in a.shape)
    if len(shape) != 3 or (shape_tail is not None and shape[1:] != tuple(shape_tail)):
        raise ValueError(f"{name}: shape {shape}, expected [N, {', '.join(map(str, shape_tail or ('*', '*')))}]")
    return _addr(a), shape


def render_sharded(build: Callable, host_in, host_out, devices: Sequence[int] = (0,), sub_batches: int = 8,
                   sample_rate: float = 48000.0, pcm16: bool = False, pull: Optional[Callable] = None, out_pcm16: bool = False,
                   reuse: bool = False):
    """Render N contexts whose source AudioBuffers live on the host, on one or several GPUs: a thin caller of the library's
    waa_render_sharded (include/waa_hip.h; csrc/waa_sharded.cpp — contiguous ranges per device, sub-batches pipelined
    upload || render || download, one host thread per sub-batch inside the library).

    build(n_instances, device) -> (ctx, src): builds the (identical) graph for a sub-batch of `n_instances` contexts on
        `device`; `src` is the AudioBufferSourceNode that receives the contexts' buffers.  Called once for the graph's shape
        and once per sub-batch (on the library's thread) to configure that sub-batch: node payloads, params, schedules.
    host_in:  [N, channels, frames] float32 — or, with pcm16=True, [N, frames, channels] int16 (decoded WAV data: half
        the upload, converted on the device) — C-contiguous numpy array or pinned torch tensor.
    host_out: [N, n_out, length] float32 (out_pcm16=True: [N, length, n_out] int16), filled with every context's AudioBuffer.
    pull(ctx, lo, hi): optional control-side work per sub-batch after its render (e.g. the batched analyser pull).
    reuse: build() configures every sub-batch identically (waa_sharded_job.reuse_batches): a downloaded sub-batch is re-armed for a
        later one of the same size instead of being destroyed — no second creation, setup or plan.
    Returns {"seconds": wall time, "shards": [(device, lo, hi), ...]}.  Raises the first sub-batch error."""
    from .api import SHARD_FN, ShardedJob, WaaError
    devices = [int(d) for d in devices]
    if not devices:
        raise ValueError("render_sharded needs at least one device")
    base_in, shape_in = _host_block(host_in, "host_in", np.int16 if pcm16 else np.float32)
    n_total = shape_in[0]
    frames, n_ch = (shape_in[1], shape_in[2]) if pcm16 else (shape_in[2], shape_in[1])
    tmpl, tsrc = build(1, devices[0])
    tail = (tmpl.length, tmpl.number_of_channels) if out_pcm16 else (tmpl.number_of_channels, tmpl.length)
    base_out, shape_out = _host_block(host_out, "host_out", np.int16 if out_pcm16 else np.float32, tail)
    if shape_out[0] != n_total:
        raise ValueError(f"host_out holds {shape_out[0]} contexts, host_in {n_total}")
    b = tmpl._b
    graph = tmpl.graph_desc()
    live, errors, by_handle = {}, [], {}

    def setup(handle, first, count, device, _user):
        try:
            ctx, _ = build(int(count), int(device))
            if len(ctx._nodes) != graph.n_nodes:
                raise WaaError(1, "render_sharded: build() must return the same graph for every sub-batch")
            live[int(first)] = ctx  # (before _adopt: whatever happens below, the finally clause forgets the library's handle)
            by_handle[int(handle)] = ctx
            ctx._adopt(handle)
            return 0
        except WaaError as e:
            errors.append(e)
            return e.status or 3
        except Exception as e:  # noqa: BLE001 — reported to the caller below
            errors.append(e)
            return 3

    def after(handle, first, count, device, _user):
        try:
            if pull is not None:  # (a re-used batch was set up for an earlier sub-batch: found by its handle)
                pull(by_handle[int(handle)], int(first), int(first) + int(count))
            return 0
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            return 3

    dev_arr = (C.c_int32 * len(devices))(*devices)
    job = ShardedJob()
    job.graph = C.pointer(graph)
    job.n_instances, job.n_channels_out, job.length_frames = n_total, tmpl.number_of_channels, tmpl.length
    job.sample_rate = tmpl.sample_rate
    job.n_devices, job.devices, job.sub_batches = len(devices), dev_arr, max(1, int(sub_batches))
    job.source_node = tsrc.id
    job.host_in, job.in_channels, job.in_pcm16, job.in_frames, job.in_sample_rate = base_in, n_ch, int(pcm16), frames, sample_rate
    job.out_pcm16, job.host_out = int(out_pcm16), base_out
    job.setup, job.pull = SHARD_FN(setup), SHARD_FN(after)
    job.reuse_batches = int(bool(reuse))
    seconds = C.c_double()
    try:
        status = b.render_sharded(C.cast(C.pointer(job), C.c_void_p), C.byref(seconds))
    finally:
        for ctx in live.values():
            ctx._release()  # (the library created and destroyed these batches)
        tmpl._release()
    if errors:
        raise errors[0]
    b.check(status)
    return {"seconds": seconds.value, "shards": [(d, lo, hi) for _, d, _, lo, hi in plan_shards(n_total, devices, sub_batches)]}
