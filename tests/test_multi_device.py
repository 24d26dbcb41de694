"""Several waa_batches driven concurrently (contexts shard over the GPUs of a node with one host
thread per device, no collective).  A batch is single-threaded; DIFFERENT batches are independent — on different
devices, or on the same one (own stream each).  The two-device case needs a node with >= 2 GPUs and skips otherwise;
the same-device case runs on the 1-GPU box and checks exactly the property the sharded run relies on: concurrent
batches do not disturb each other and the union of the shards equals the unsharded render."""
import threading

import numpy as np
import pytest

import web_audio_api_rs_amd as waa
from graphs import c2, rms_err, white_noise
from web_audio_api_rs_amd.sharding import shard_range


def _render_shards(hip, noise, devices):
    n = noise.shape[0]
    world = len(devices)
    outs, errs = [None] * world, []

    def work(rank):
        try:
            lo, hi = shard_range(n, rank, world)
            ctx, _ = c2(hip, noise[lo:hi], device=devices[rank])
            outs[rank] = ctx.start_rendering_sync().data
            ctx.close()
        except Exception as e:  # surfaced below: a failing thread must fail the test
            errs.append(e)

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    return np.concatenate(outs, axis=0)


@pytest.mark.gpu
def test_two_batches_on_two_devices(hip, orc):
    if hip.device_count() < 2:
        pytest.skip("needs a node with at least two GPUs (the driver's scaling run covers it otherwise)")
    noise = white_noise(10, 2, 128 * 300 + 7)
    got = _render_shards(hip, noise, [0, 1])
    ctx, _ = c2(orc, noise)
    ref = ctx.start_rendering_sync().data
    ctx.close()
    assert rms_err(got, ref).max() <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_concurrent_batches_on_one_device(hip, orc, world):
    noise = white_noise(11, 2, 128 * 300 + 7)
    got = _render_shards(hip, noise, [0] * world)
    ctx, _ = c2(hip, noise, device=0)
    whole = ctx.start_rendering_sync().data
    ctx.close()
    assert np.array_equal(got, whole)  # sharding changes nothing, bit for bit
    ctx, _ = c2(orc, noise)
    ref = ctx.start_rendering_sync().data
    ctx.close()
    assert rms_err(got, ref).max() <= 1e-6


@pytest.mark.gpu
def test_device_arena_serves_the_big_buffers_and_changes_nothing(hip, orc):
    """waa_device_arena_reserve: batches created afterwards carve their large buffers out of one slab (2 MB-aligned pieces,
    handed back when the last batch using the slab is destroyed); results are bit-identical to plain hipMalloc, a batch that
    does not fit falls back to hipMalloc, releasing a slab in use is an InvalidStateError"""
    from graphs import c2, white_noise
    noise = white_noise(6, 2, 128 * 300)

    def render():
        ctx, _ = c2(hip, noise)
        out = ctx.start_rendering_sync().data
        ptr = ctx.output_device()[0]
        return ctx, out, ptr

    ctx, plain, _ = render()
    ctx.close()
    hip.check(hip.device_arena_reserve(0, 64 << 20))
    try:
        ctx, out, ptr = render()
        assert np.array_equal(out, plain)
        assert ptr % (2 << 20) == 0           # carved from the slab
        ctx2, out2, ptr2 = render()           # a second batch next to the first
        assert np.array_equal(out2, plain) and ptr2 != ptr and ptr2 % (2 << 20) == 0
        with pytest.raises(waa.WaaError, match="InvalidStateError"):
            hip.check(hip.device_arena_reserve(0, 0))
        ctx.close()
        ctx2.close()
        hip.check(hip.device_arena_reserve(0, 1 << 20))   # too small for anything: everything falls back
        ctx, out, _ = render()
        assert np.array_equal(out, plain)
        ctx.close()
    finally:
        hip.check(hip.device_arena_reserve(0, 0))


@pytest.mark.gpu
def test_device_arena_hands_pieces_back_while_lifetimes_overlap(hip):
    """the bump allocator rewound only when NO piece was live — a pipeline (the next batch created before the
    previous one is destroyed) ran the slab dry and every later batch fell back to hipMalloc without a sign.  The free list
    returns a batch's pieces when THAT batch is destroyed: 40 overlapping batches through a slab that holds about three,
    zero misses (waa_device_arena_stats), and the caller's current device is left alone."""
    import torch
    from graphs import c2, white_noise
    noise = white_noise(4, 2, 128 * 2000)  # 2 MB per plane set: source + output pieces of a few MB per batch
    before = torch.cuda.current_device()
    hip.check(hip.device_arena_reserve(0, 96 << 20))
    assert torch.cuda.current_device() == before
    try:
        st = waa.arena_stats(hip, 0)
        assert st["reserved_bytes"] == 96 << 20 and st["in_use_bytes"] == 0 and st["misses"] == 0
        prev = None
        first = None
        for i in range(40):
            ctx, _ = c2(hip, noise, device=0)
            out = ctx.start_rendering_sync().data
            first = out if first is None else first
            assert np.array_equal(out, first)
            if prev is not None:
                prev.close()
            prev = ctx
        st = waa.arena_stats(hip, 0)
        assert st["misses"] == 0 and st["served"] >= 80, st
        assert 0 < st["in_use_bytes"] <= st["peak_bytes"] <= 96 << 20
        prev.close()
        st = waa.arena_stats(hip, 0)
        assert st["in_use_bytes"] == 0 and st["largest_free_bytes"] == 96 << 20, st  # everything merged back into one block
        # a request the slab cannot serve is counted
        big = white_noise(4, 2, 128 * 40000)
        ctx, _ = c2(hip, big, device=0)
        ctx.prepare()
        assert waa.arena_stats(hip, 0)["misses"] >= 1
        ctx.close()
    finally:
        hip.check(hip.device_arena_reserve(0, 0))
    assert waa.arena_stats(hip, 0)["reserved_bytes"] == 0


@pytest.mark.gpu
def test_graded_arena_sorts_its_units_and_serves_reads_from_the_top(hip, orc):
    """waa_device_arena_reserve_graded (round 6): candidate physical units are timed as the destination of C2's copy shape, the
    fastest are mapped side by side in ascending order of that time, the rest released; a batch's written buffers come from the
    bottom of the slab, its source AudioBuffers (only read) from the top; the render is bit-identical to plain hipMalloc."""
    from graphs import c2, white_noise
    noise = white_noise(6, 2, 128 * 300)

    def render():
        ctx, _ = c2(hip, noise)
        out = ctx.start_rendering_sync().data
        return ctx, out, ctx.output_device()[0]

    ctx, plain, _ = render()
    ctx.close()
    free0 = __import__("torch").cuda.mem_get_info()[0]
    hip.check(hip.device_arena_reserve_graded(0, 256 << 20, 1024 << 20))
    try:
        g = waa.arena_grades(hip, 0)
        assert g["unit_bytes"] == 64 << 20 and g["n_units"] == 4 and g["n_candidates"] == 16, g
        assert g["unit_ms"] == sorted(g["unit_ms"]) and 0 < g["best_ms"] == g["unit_ms"][0] <= g["worst_kept_ms"] <= g["worst_candidate_ms"]
        st = waa.arena_stats(hip, 0)
        assert st["reserved_bytes"] == 256 << 20 and st["in_use_bytes"] == 0
        # the surplus candidates went back to the device
        assert free0 - __import__("torch").cuda.mem_get_info()[0] < (256 + 64) << 20
        ctx, out, ptr = render()
        assert np.array_equal(out, plain)
        ctx2, out2, ptr2 = render()
        assert np.array_equal(out2, plain) and ptr2 != ptr
        with pytest.raises(waa.WaaError, match="InvalidStateError"):
            hip.check(hip.device_arena_reserve_graded(0, 0, 0))
        ctx.close()
        ctx2.close()
        assert waa.arena_stats(hip, 0)["in_use_bytes"] == 0
    finally:
        hip.check(hip.device_arena_reserve(0, 0))
    assert waa.arena_grades(hip, 0)["n_units"] == 0 and waa.arena_stats(hip, 0)["reserved_bytes"] == 0
    # big buffers: the output below the source (white_noise 2 MB+ planes are carved from the slab: >= 1 MB)
    big = white_noise(8, 2, 128 * 4000)
    hip.check(hip.device_arena_reserve_graded(0, 512 << 20, 512 << 20))
    try:
        ctx, _ = c2(hip, big)
        out = ctx.start_rendering_sync().data
        st = waa.arena_stats(hip, 0)
        assert st["served"] >= 2 and st["misses"] == 0
        ctx.close()
    finally:
        hip.check(hip.device_arena_reserve(0, 0))
    ctx, _ = c2(orc, big)
    ref = ctx.start_rendering_sync().data
    ctx.close()
    assert rms_err(out, ref).max() <= 1e-6
