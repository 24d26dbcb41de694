"""sharding.render_sharded — the N-device render component: host buffers of all contexts ->
contiguous ranges per device -> pipelined sub-batches (upload || render || download) -> host buffers.  The logic is
backend independent, so the partition / ordering / error paths are exercised here with the CPU oracle standing in for the
device library; the GPU tests run it on the HIP library (several slots on one device on a 1-GPU box, two devices where
there are two) and require the union of the shards to equal the unsharded render bit for bit."""
import ctypes as C

import numpy as np
import pytest

import web_audio_api_rs_amd as waa
from graphs import c2, rms_err, white_noise
from web_audio_api_rs_amd.sharding import plan_shards, render_sharded

RQ = 128


def _build(be, with_analyser=False):
    def build(n, device):
        ctx = waa.OfflineAudioContext(2, RQ * 30 + 5, 48000.0, n_instances=n, binding=be, device=device)
        src = ctx.create_buffer_source()
        node = src.connect(ctx.create_biquad_filter(type_="lowpass", frequency=200.0, q=1.0)).connect(ctx.create_gain(gain=0.5))
        if with_analyser:
            node = node.connect(ctx.create_analyser(fft_size=256))
        node.connect(ctx.destination())
        src.start()
        return ctx, src
    return build


def test_plan_shards_covers_everything_in_order():
    for n in (1, 5, 8, 64, 513):
        for devices in ([0], [0, 1], [0, 0, 1], list(range(8))):
            for parts in (1, 3, 8):
                sh = plan_shards(n, devices, parts)
                cover = sorted((lo, hi) for _, _, _, lo, hi in sh)
                assert cover[0][0] == 0 and cover[-1][1] == n
                assert all(cover[i][1] == cover[i + 1][0] for i in range(len(cover) - 1))
                for slot in range(len(devices)):
                    ks = [k for s, _, k, _, _ in sh if s == slot]
                    assert ks == list(range(len(ks)))  # the turn-taking order of a slot has no holes


@pytest.mark.parametrize("devices,parts", [([-1], 1), ([-1], 3), ([-1, -1], 2), ([-1, -1, -1], 8)])
def test_sharded_render_equals_single_batch_oracle(orc, devices, parts):
    n, frames = 7, RQ * 30 + 5
    noise = white_noise(n, 2, frames)
    out = np.zeros((n, 2, frames), np.float32)
    info = render_sharded(_build(orc), noise, out, devices=devices, sub_batches=parts)
    assert sorted((lo, hi) for _, lo, hi in info["shards"])[0][0] == 0 and info["seconds"] > 0
    ctx, src = _build(orc)(n, -1)
    src.set_buffer_batch(noise, 48000.0)
    ref = ctx.start_rendering_sync().data
    ctx.close()
    assert np.array_equal(out, ref)


def test_sharded_render_pcm16_and_pull(orc):
    n, frames = 5, RQ * 30 + 5
    rng = np.random.default_rng(3)
    pcm = rng.integers(-32768, 32767, (n, frames, 2), dtype=np.int16)
    out = np.zeros((n, 2, frames), np.float32)
    bins = np.zeros((n, 128), np.float32)

    def pull(ctx, lo, hi):
        an = next(nd for nd in ctx._nodes if isinstance(nd, waa.AnalyserNode))
        an.get_float_frequency_data_all(out=bins[lo:hi])

    render_sharded(_build(orc, with_analyser=True), pcm, out, devices=[-1, -1], sub_batches=2, pcm16=True, pull=pull)
    ctx, src = _build(orc, with_analyser=True)(n, -1)
    src.set_buffer_batch(np.ascontiguousarray(pcm.transpose(0, 2, 1).astype(np.float32) / np.float32(32768.0)), 48000.0)
    ref = ctx.start_rendering_sync().data
    an = next(nd for nd in ctx._nodes if isinstance(nd, waa.AnalyserNode))
    ref_bins = an.get_float_frequency_data_all()
    ctx.close()
    assert np.array_equal(out, ref) and np.array_equal(bins, ref_bins)


def test_a_failing_sub_batch_raises_and_does_not_hang(orc):
    calls = []

    def build(n, device):
        calls.append(n)
        if len(calls) == 3:  # (the template, the first sub-batch, then the second of the two: 5 contexts = 3 + 2)
            raise waa.WaaError(2, "NotSupportedError - the second sub-batch refuses its configuration")
        return _build(orc)(n, device)
    noise = white_noise(5, 2, RQ * 30 + 5)
    with pytest.raises(waa.WaaError, match="NotSupportedError"):
        render_sharded(build, noise, np.zeros_like(noise), devices=[-1], sub_batches=2)


def test_a_node_that_fails_inside_setup_leaves_no_dangling_handle(orc):
    """when a node's _apply raises INSIDE the setup callback (after the context adopted the
    library's batch), the context must forget the handle — waa_render_sharded destroys the batch itself, and a context that
    still held it destroyed it a second time from __del__ (double free).  The contexts are kept and closed explicitly here."""
    import gc
    made = []

    def build(n, device):
        ctx, src = _build(orc)(n, device)
        made.append(ctx)
        if len(made) == 3:  # (the template, the first sub-batch, then the second)
            gain = next(nd for nd in ctx._nodes if isinstance(nd, waa.GainNode))
            orig = gain._apply

            def refuse(c):
                orig(c)
                raise waa.WaaError(2, "NotSupportedError - this node refuses its payload inside setup")
            gain._apply = refuse
        return ctx, src
    noise = white_noise(5, 2, RQ * 30 + 5)
    with pytest.raises(waa.WaaError, match="refuses its payload"):
        render_sharded(build, noise, np.zeros_like(noise), devices=[-1], sub_batches=2)
    assert len(made) == 3
    for ctx in made:
        assert ctx._handle is None  # nobody but the library owns a batch any more
        ctx.close()
    del made
    gc.collect()
    # the library is still healthy: the same job without the refusal renders
    out = np.zeros_like(noise)
    render_sharded(_build(orc), noise, out, devices=[-1], sub_batches=2)
    assert np.abs(out).max() > 1e-3


def test_close_never_destroys_an_adopted_batch(orc):
    ctx, _ = _build(orc)(1, -1)
    calls = []
    ctx._b = type("B", (), {"batch_destroy": lambda self, h: calls.append(h)})()
    ctx._handle, ctx._foreign = object(), True
    ctx.close()
    assert calls == [] and ctx._handle is None
    # ... and a batch the context creates for ITSELF afterwards is its own again (it leaked: _foreign stayed set)
    assert ctx._foreign is False
    own = object()
    ctx._handle = own
    ctx.close()
    assert calls == [own]
    ctx._handle, ctx._foreign = object(), True
    ctx._release()
    assert ctx._foreign is False and ctx._handle is None


@pytest.mark.gpu
@pytest.mark.parametrize("slots,parts", [(1, 4), (2, 2), (3, 3)])
def test_sharded_render_equals_single_batch_hip(hip, orc, slots, parts):
    n, frames = 13, RQ * 30 + 5
    noise = white_noise(n, 2, frames)
    out = np.zeros((n, 2, frames), np.float32)
    ndev = hip.device_count()
    devices = [s % ndev for s in range(slots)]  # (distinct devices where the box has them)
    render_sharded(_build(hip), noise, out, devices=devices, sub_batches=parts)
    ctx, src = _build(hip)(n, 0)
    src.set_buffer_batch(noise, 48000.0)
    whole = ctx.start_rendering_sync().data
    ctx.close()
    assert np.array_equal(out, whole)  # sharding changes nothing, bit for bit
    ctx, src = _build(orc)(n, -1)
    src.set_buffer_batch(noise, 48000.0)
    ref = ctx.start_rendering_sync().data
    ctx.close()
    assert rms_err(out, ref).max() <= 1e-6


@pytest.mark.gpu
def test_sharded_render_two_devices_hip(hip):
    if hip.device_count() < 2:
        pytest.skip("needs a node with at least two GPUs (the driver's scaling run covers it otherwise)")
    n, frames = 16, RQ * 30 + 5
    noise = white_noise(n, 2, frames)
    out = np.zeros((n, 2, frames), np.float32)
    info = render_sharded(_build(hip), noise, out, devices=[0, 1], sub_batches=2)
    assert {d for d, _, _ in info["shards"]} == {0, 1}
    ctx, src = _build(hip)(n, 0)
    src.set_buffer_batch(noise, 48000.0)
    whole = ctx.start_rendering_sync().data
    ctx.close()
    assert np.array_equal(out, whole)


def test_sharded_render_pcm16_output(orc):
    """out_pcm16: every context's AudioBuffer as interleaved 16-bit PCM (waa_download_all_pcm16): sample * 32768 rounded to
    nearest and saturated — half the bytes back over the link"""
    n, frames = 5, RQ * 30 + 5
    noise = white_noise(n, 2, frames) * 1.9   # (some samples beyond +-1 after the gain of 0.5? no: keep a few that saturate)
    noise[0, 0, :8] = [4.0, -4.0, 2.1, -2.1, 0.0, 1e-6, -1e-6, 1.999]
    out16 = np.zeros((n, frames, 2), np.int16)
    render_sharded(_build(orc), noise, out16, devices=[-1, -1], sub_batches=2, out_pcm16=True)
    out = np.zeros((n, 2, frames), np.float32)
    render_sharded(_build(orc), noise, out, devices=[-1], sub_batches=1)
    want = np.clip(np.rint(out.astype(np.float64) * 32768.0), -32768, 32767).astype(np.int16).transpose(0, 2, 1)
    assert np.array_equal(out16, want)


def test_host_buffers_are_validated():
    be = waa.default_binding()
    build = _build(be)
    good_in, good_out = np.zeros((3, 2, RQ * 30 + 5), np.float32), np.zeros((3, 2, RQ * 30 + 5), np.float32)
    with pytest.raises(ValueError, match="host_in"):
        render_sharded(build, good_in.astype(np.float64), good_out, devices=[waa.PLAN_ONLY])
    with pytest.raises(ValueError, match="host_in"):
        render_sharded(build, good_in[:, :, ::2], good_out, devices=[waa.PLAN_ONLY])
    with pytest.raises(ValueError, match="host_out"):
        render_sharded(build, good_in, np.zeros((3, 2, RQ * 30), np.float32), devices=[waa.PLAN_ONLY])
    with pytest.raises(ValueError, match="contexts"):
        render_sharded(build, good_in, np.zeros((4, 2, RQ * 30 + 5), np.float32), devices=[waa.PLAN_ONLY])


@pytest.mark.gpu
def test_sharded_render_pcm16_output_hip(hip, orc):
    n, frames = 9, RQ * 30 + 5
    noise = white_noise(n, 2, frames)
    out16 = np.zeros((n, frames, 2), np.int16)
    render_sharded(_build(hip), noise, out16, devices=[0], sub_batches=3, out_pcm16=True)
    ref16 = np.zeros((n, frames, 2), np.int16)
    render_sharded(_build(orc), noise, ref16, devices=[-1], sub_batches=1, out_pcm16=True)
    assert np.abs(out16.astype(int) - ref16.astype(int)).max() <= 1   # (the f32 renders agree to 1e-7: at most one code apart)
    assert np.abs(out16).max() > 1000


def _build_heavy(be, ir):
    """source -> Convolver(IR longer than 64 K frames: the 160 KB-LDS transforms) -> Analyser(32768: the 128 KB analyser
    workgroup) -> destination: the kernels whose dynamic-LDS limit has to be raised on EVERY device they run on"""
    def build(n, device):
        ctx = waa.OfflineAudioContext(2, 8192 * 5 + 300, 48000.0, n_instances=n, binding=be, device=device)
        src = ctx.create_buffer_source()
        node = src.connect(ctx.create_convolver(buffer=waa.AudioBuffer(ir, 48000.0))).connect(ctx.create_analyser(fft_size=32768))
        node.connect(ctx.destination())
        src.start()
        return ctx, src
    return build


@pytest.mark.gpu
@pytest.mark.parametrize("two_devices", [False, True])
def test_sharded_render_with_the_big_lds_kernels(hip, orc, two_devices):
    """the >64 KB dynamic-LDS limit of a kernel is a per-device attribute; it used to be raised once per
    PROCESS (a static flag), so a second device driven from the same process would have failed its first long-IR convolver
    or 32768-point analyser launch.  One device, two slots: runs everywhere; two devices: where the box has them."""
    if two_devices and hip.device_count() < 2:
        pytest.skip("needs a node with at least two GPUs")
    from graphs import garage_like_ir
    ir = garage_like_ir(frames=70000)
    n, frames = 6, 8192 * 5 + 300
    noise = white_noise(n, 2, frames)
    out = np.zeros((n, 2, frames), np.float32)
    bins = np.zeros((n, 16384), np.float32)

    def pull(ctx, lo, hi):
        an = next(nd for nd in ctx._nodes if isinstance(nd, waa.AnalyserNode))
        an.get_float_frequency_data_all(out=bins[lo:hi])

    render_sharded(_build_heavy(hip, ir), noise, out, devices=[0, 1] if two_devices else [0, 0], sub_batches=2, pull=pull)
    ctx, src = _build_heavy(orc, ir)(n, -1)
    src.set_buffer_batch(noise, 48000.0)
    ref = ctx.start_rendering_sync().data
    an = next(nd for nd in ctx._nodes if isinstance(nd, waa.AnalyserNode))
    ref_bins = an.get_float_frequency_data_all()
    ctx.close()
    assert rms_err(out, ref).max() <= 1e-6 and np.abs(ref).max() > 1e-3
    gl, ol = 10.0 ** (bins.astype(np.float64) / 20), 10.0 ** (ref_bins.astype(np.float64) / 20)
    assert (np.abs(gl - ol).max(axis=1) / ol.max(axis=1)).max() <= 2e-5   # (a 32768-point f32 transform on each side)


def _build_modulated(be):
    """an LFO on the Biquad's frequency and a vibrato on the streamed source's playbackRate: the second makes the plan render the
    modulating subgraph (waa_render_sharded must then plan AFTER the source's data is there, not before its turn on the link)"""
    def build(n, device):
        ctx = waa.OfflineAudioContext(2, RQ * 40 + 5, 48000.0, n_instances=n, binding=be, device=device)
        src = ctx.create_buffer_source()
        bq = ctx.create_biquad_filter(type_="lowpass", frequency=900.0, q=1.0)
        lfo = ctx.create_oscillator(type_="sine", frequency=5.0)
        lfo.connect(ctx.create_gain(gain=300.0)).connect(bq.frequency)
        vib = ctx.create_oscillator(type_="sine", frequency=3.0)
        vib.connect(ctx.create_gain(gain=0.02)).connect(src.playback_rate)
        src.connect(bq).connect(ctx.destination())
        lfo.start()
        vib.start()
        src.start()
        return ctx, src
    return build


@pytest.mark.gpu
def test_sharded_render_of_a_graph_with_modulated_params_hip(hip, orc):
    n, frames = 7, RQ * 40 + 5
    noise = white_noise(n, 2, frames)
    out = np.zeros((n, 2, frames), np.float32)
    render_sharded(_build_modulated(hip), noise, out, devices=[0], sub_batches=3)
    ctx, src = _build_modulated(hip)(n, 0)
    src.set_buffer_batch(noise, 48000.0)
    whole = ctx.start_rendering_sync().data
    ctx.close()
    assert np.array_equal(out, whole)
    ctx, src = _build_modulated(orc)(n, -1)
    src.set_buffer_batch(noise, 48000.0)
    ref = ctx.start_rendering_sync().data
    ctx.close()
    assert rms_err(out, ref).max() <= 1e-6 and np.abs(ref).max() > 0.05


@pytest.mark.gpu
def test_sharded_render_pcm16_input_at_another_rate_hip(hip, orc):
    """16-bit PCM at 44.1 kHz into a 48 kHz context: the deferred fill goes through the device's decode + resample kernel"""
    n, frames_in = 9, 3600
    rng = np.random.default_rng(5)
    pcm = rng.integers(-20000, 20000, (n, frames_in, 2)).astype(np.int16)
    out = np.zeros((n, 2, RQ * 30 + 5), np.float32)
    render_sharded(_build(hip), pcm, out, devices=[0], sub_batches=4, pcm16=True, sample_rate=44100.0)
    ref = np.zeros_like(out)
    render_sharded(_build(orc), pcm, ref, devices=[-1], sub_batches=1, pcm16=True, sample_rate=44100.0)
    assert rms_err(out, ref).max() <= 1e-6 and np.abs(ref).max() > 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("window", [1, 2, 0])
def test_sharded_render_with_a_bounded_number_of_sub_batches_in_flight(hip, window):
    """waa_sharded_in_flight: at most `window` sub-batches of a device exist at a time — 1 serialises them, 0 is
    the unbounded behaviour of round 4; the result never changes and nothing deadlocks (8 sub-batches through every window)."""
    n, frames = 24, RQ * 30 + 5
    noise = white_noise(n, 2, frames)
    ref = np.zeros((n, 2, frames), np.float32)
    render_sharded(_build(hip), noise, ref, devices=[0], sub_batches=1)
    out = np.zeros_like(ref)
    hip.check(hip.sharded_in_flight(window))
    try:
        render_sharded(_build(hip), noise, out, devices=[0, 0], sub_batches=8)
    finally:
        hip.check(hip.sharded_in_flight(4))
    assert np.array_equal(out, ref)


@pytest.mark.gpu
def test_in_flight_window_bounds_device_memory(hip):
    """with window 1 the arena's peak holds ONE sub-batch's pieces, with no bound all eight"""
    n, frames = 16, RQ * 4000
    noise = white_noise(n, 2, frames)
    out = np.zeros_like(noise)

    def build(k, device):
        ctx = waa.OfflineAudioContext(2, frames, 48000.0, n_instances=k, binding=hip, device=device)
        src = ctx.create_buffer_source()
        src.connect(ctx.create_biquad_filter(type_="lowpass", frequency=200.0, q=1.0)).connect(ctx.destination())
        src.start()
        return ctx, src
    peaks = {}
    for window in (1, 0):
        hip.check(hip.device_arena_reserve(0, 512 << 20))
        hip.check(hip.sharded_in_flight(window))
        try:
            render_sharded(build, noise, out, devices=[0], sub_batches=8)
            st = waa.arena_stats(hip, 0)
            assert st["misses"] == 0 and st["in_use_bytes"] == 0
            peaks[window] = st["peak_bytes"]
        finally:
            hip.check(hip.sharded_in_flight(4))
            hip.check(hip.device_arena_reserve(0, 0))
    assert peaks[1] * 3 <= peaks[0], peaks


@pytest.mark.gpu
def test_pcm16_download_of_more_contexts_than_grid_rows(hip, orc):
    """the PCM packer put the context index on gridDim.y (65535 rows at most) — a batch of 70 000 short contexts
    failed at the launch; it now walks the rows in strides"""
    n, frames = 70000, 64
    rng = np.random.default_rng(3)
    noise = rng.uniform(-1, 1, (n, 1, frames)).astype(np.float32)
    outs = []
    for be in (hip, orc):
        ctx = waa.OfflineAudioContext(1, frames, 48000.0, n_instances=n, binding=be)
        src = ctx.create_buffer_source()
        src.set_buffer_batch(noise, 48000.0)
        src.connect(ctx.destination())
        src.start()
        ctx.render_async()
        pcm = np.empty((n, frames, 1), np.int16)
        be.check(be.download_all_pcm16(ctx._handle, pcm.ctypes.data_as(C.POINTER(C.c_int16))))
        ctx.close()
        outs.append(pcm)
    assert np.array_equal(outs[0], outs[1]) and np.abs(outs[0][-1].astype(np.int32)).max() > 100


@pytest.mark.gpu
def test_rearmed_batch_renders_new_audio_without_a_new_plan(hip, orc):
    """waa_batch_rearm (round 6): a planned batch takes the next AudioBuffers of the same shape into the device buffers it was
    planned with; the render is the one a fresh batch gives, bit for bit; another shape is an InvalidStateError"""
    n, frames = 6, RQ * 30 + 5
    a, b2 = white_noise(n, 2, frames), white_noise(n, 2, frames, seed0=77)
    ctx, src = _build(hip)(n, 0)
    src.set_buffer_batch(a, 48000.0)
    first = ctx.start_rendering_sync().data
    hb, h = ctx._b, ctx._handle
    hb.check(hb.batch_rearm(h))
    with pytest.raises(waa.WaaError, match="InvalidStateError"):   # nothing rendered (yet) after a re-arm
        hb.check(hb.download_all(h, waa.api._fp(np.empty_like(first))))
    with pytest.raises(waa.WaaError, match="the shape it was planned with"):
        hb.check(hb.source_set_buffer_batch(h, src.id, waa.api._fp(b2[:, :, :-1].copy()), 2, frames - 1, 48000.0))
    hb.check(hb.source_set_buffer_batch(h, src.id, waa.api._fp(b2), 2, frames, 48000.0))
    hb.check(hb.render(h))
    second = np.empty_like(first)
    hb.check(hb.download_all(h, waa.api._fp(second)))
    ctx.close()
    fresh, fsrc = _build(hip)(n, 0)
    fsrc.set_buffer_batch(b2, 48000.0)
    want = fresh.start_rendering_sync().data
    fresh.close()
    assert np.array_equal(second, want) and not np.array_equal(second, first)


@pytest.mark.gpu
def test_sharded_render_with_reused_batches_equals_the_plain_pipeline(hip):
    """waa_sharded_job.reuse_batches: later sub-batches of the same size re-arm a downloaded one — same bits"""
    n, frames = 32, RQ * 30 + 5
    noise = white_noise(n, 2, frames)
    outs = []
    for reuse in (False, True):
        out = np.zeros((n, 2, frames), np.float32)
        render_sharded(_build(hip), noise, out, devices=[0], sub_batches=8, reuse=reuse)
        outs.append(out)
    assert np.array_equal(outs[0], outs[1]) and np.abs(outs[0]).max() > 1e-3


def test_sharded_render_reuse_flag_on_the_oracle(orc):
    n, frames = 9, RQ * 30 + 5
    noise = white_noise(n, 2, frames)
    outs = []
    for reuse in (False, True):
        out = np.zeros((n, 2, frames), np.float32)
        render_sharded(_build(orc), noise, out, devices=[-1], sub_batches=3, reuse=reuse)
        outs.append(out)
    assert np.array_equal(outs[0], outs[1])
