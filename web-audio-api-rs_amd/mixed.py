"""Many independent OfflineAudioContexts of DIFFERENT shapes -> as few device batches as their shapes allow.

The reference's unit is the context (src/context/offline.rs:78-143): a host that serves requests builds one
OfflineAudioContext per request, each with its own graph.  The library's unit is the batch — N contexts of ONE graph shape
rendered together (include/waa_hip.h) — so a serving host has to sort its contexts by shape first.  `render_contexts` is that
step for the Python mirror (round-5 review, missing 4): contexts built one by one, exactly like reference contexts
(`OfflineAudioContext(channels, length, rate)` with n_instances = 1), are bucketed by `shape_key` — the graph (node kinds, options,
channel configuration, connections), the context's format, the SHAPES of per-context payloads (an AudioBufferSource's channel count,
length and rate) and the identity of payloads a batch shares (a convolver's impulse response, a shaper's curve, IIR coefficients,
a periodic wave) — every bucket is merged into one batch (per-context AudioBuffers, AudioParam values and automation, start / stop /
loop settings become per-instance settings of the batch), rendered, and the AudioBuffers are handed back in the callers' order.
Contexts with suspend callbacks render on their own (their callbacks may do anything)."""
from __future__ import annotations

import hashlib
from typing import List, Sequence

import numpy as np

from .api import ALL, AudioBufferSourceNode, AudioParam, ConvolverNode, IIRFilterNode, OfflineAudioContext, OscillatorNode, RenderedBatch, \
    WaveShaperNode, _ScheduledSource


def _digest(a) -> str:
    a = np.ascontiguousarray(a)
    return hashlib.sha1(a.tobytes()).hexdigest()[:16] + str(a.shape) + str(a.dtype)


def _node_key(nd) -> tuple:
    d = nd._desc()
    key = [type(nd).__name__, int(d.kind), int(d.channel_count), int(d.channel_count_mode), int(d.channel_interpretation),
           tuple(int(x) for x in d.i), tuple(float(x) for x in d.d)]
    if isinstance(nd, ConvolverNode):  # (shared by the batch: another response is another batch)
        key.append(None if nd.buffer is None else (_digest(nd.buffer.data), float(nd.buffer.sample_rate)))
        pcm = getattr(nd, "_pcm", None)
        key.append(None if pcm is None else (_digest(pcm[0]), pcm[1]))
    if isinstance(nd, WaveShaperNode):
        curve = getattr(nd, "curve", None)
        key.append(None if curve is None else _digest(curve))
    if isinstance(nd, IIRFilterNode):
        key.append((_digest(nd.feedforward), _digest(nd.feedback)))
    if isinstance(nd, OscillatorNode) and nd.periodic_wave is not None:
        w = nd.periodic_wave
        key.append(_digest(w.table) if w.table is not None else (_digest(w.real), _digest(w.imag), w.disable_normalization))
    if isinstance(nd, AudioBufferSourceNode):
        # per-context payloads: their SHAPE is part of the batch's shape (one device buffer [instance][channel][frame])
        if nd._batch is not None or nd._pcm is not None or nd._pcm_one or nd._device is not None:
            key.append(("own-upload", id(nd)))  # (pre-batched / PCM / device-resident payloads: not merged)
        buf = nd._buffers.get(ALL)
        key.append(None if buf is None else (buf.number_of_channels, buf.length, float(buf.sample_rate)))
    return tuple(key)


def shape_key(ctx: OfflineAudioContext) -> tuple:
    """what two contexts must agree on to be rendered as two instances of one batch"""
    if ctx.n_instances != 1 or ctx._suspends or ctx._ctl or ctx._handle is not None:
        return ("alone", id(ctx))
    return (ctx.number_of_channels, ctx.length, ctx.sample_rate, ctx.device, id(ctx._b), tuple(_node_key(nd) for nd in ctx._nodes),
            tuple(ctx._edges))


def _merge_param(dst: AudioParam, src: AudioParam, i: int):
    dst._const[i] = src._const[ALL]
    dst._blocks += [(q0, v, i) for (q0, v, inst) in src._blocks]
    dst._events += [(k, v, t, a, c, i) for (k, v, t, a, c, inst) in src._events]


def _own_instance_zero(p: AudioParam):
    """the carrier's own blocks and events, so far "for all", are instance 0's"""
    p._blocks = [(q0, v, 0 if inst == ALL else inst) for (q0, v, inst) in p._blocks]
    p._events = [(k, v, t, a, c, 0 if inst == ALL else inst) for (k, v, t, a, c, inst) in p._events]


def _merge(bucket: Sequence[OfflineAudioContext]) -> OfflineAudioContext:
    """bucket[0] becomes the batch (n_instances = len(bucket)); the others' per-context settings become per-instance settings"""
    car = bucket[0]
    n = len(bucket)
    if n == 1:
        return car
    car.n_instances = n
    # anything every context sets identically stays a setting "for all" (automation shared by the batch is evaluated once)
    def same(get):
        first = get(car)
        return all(get(o) == first for o in bucket[1:])
    for k, cn in enumerate(car._nodes):
        others = [o._nodes[k] for o in bucket[1:]]
        for pi, p in enumerate(cn.params):
            ops = [o.params[pi] for o in others]
            ev = lambda q: [(a, b, c, d, None if e is None else e.tobytes()) for (a, b, c, d, e, _) in q._events]  # noqa: E731
            bl = lambda q: [(a, b.tobytes()) for (a, b, _) in q._blocks]  # noqa: E731
            if all(ev(o) == ev(p) and bl(o) == bl(p) for o in ops):
                for i, o in enumerate(ops, start=1):
                    if o._const[ALL] != p._const[ALL]:
                        p._const[i] = o._const[ALL]
                continue
            _own_instance_zero(p)
            for i, o in enumerate(ops, start=1):
                _merge_param(p, o, i)
        if isinstance(cn, _ScheduledSource):
            if not all(o._starts == cn._starts and o._stops == cn._stops for o in others):
                cn._starts = {(0 if inst == ALL else inst): v for inst, v in cn._starts.items()}
                cn._stops = {(0 if inst == ALL else inst): v for inst, v in cn._stops.items()}
                for i, o in enumerate(others, start=1):
                    if ALL in o._starts:
                        cn._starts[i] = o._starts[ALL]
                    if ALL in o._stops:
                        cn._stops[i] = o._stops[ALL]
        if isinstance(cn, AudioBufferSourceNode):
            if not all(o._buffers.get(ALL) is cn._buffers.get(ALL) for o in others):
                if ALL in cn._buffers:
                    cn._buffers = {0: cn._buffers[ALL]}
                for i, o in enumerate(others, start=1):
                    if ALL in o._buffers:
                        cn._buffers[i] = o._buffers[ALL]
            if not all(o._loop == cn._loop for o in others):
                cn._loop = {(0 if inst == ALL else inst): v for inst, v in cn._loop.items()}
                for i, o in enumerate(others, start=1):
                    if ALL in o._loop:
                        cn._loop[i] = o._loop[ALL]
    for pi, p in enumerate(car._listener.params):  # the AudioListener's nine params (addressed through the panners)
        ops = [o._listener.params[pi] for o in bucket[1:]]
        if any(o._events or o._blocks or o._const[ALL] != p._const[ALL] for o in ops) or p._events or p._blocks:
            _own_instance_zero(p)
            for i, o in enumerate(ops, start=1):
                _merge_param(p, o, i)
    return car


def render_contexts(contexts: Sequence[OfflineAudioContext]) -> List[RenderedBatch]:
    """start_rendering_sync of every context, as few device batches as their shapes allow; results in the callers' order (each a
    RenderedBatch of one instance).  The contexts are consumed (like a reference context by its render): the first context of a
    bucket becomes the batch and is closed afterwards."""
    buckets = {}
    for idx, ctx in enumerate(contexts):
        buckets.setdefault(shape_key(ctx), []).append(idx)
    out: List[RenderedBatch] = [None] * len(contexts)  # type: ignore[list-item]
    for key, members in buckets.items():
        batch = _merge([contexts[i] for i in members])
        data = batch.start_rendering_sync().data
        for slot, i in enumerate(members):
            out[i] = RenderedBatch(data[slot:slot + 1].copy(), batch.sample_rate)
        batch.close()
    return out


def bucket_report(contexts: Sequence[OfflineAudioContext]) -> List[List[int]]:
    """which contexts would share a batch (indices, in first-seen order of their shapes): for logs and tests"""
    buckets = {}
    for idx, ctx in enumerate(contexts):
        buckets.setdefault(shape_key(ctx), []).append(idx)
    return list(buckets.values())
