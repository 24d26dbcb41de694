"""Graph cycles: the cycle breaker of a DelayNode (src/render/graph.rs:323-487,
src/node/delay.rs:361,535-541,693-701) and muted cycles.  Reference tests re-typed: tests/offline.rs:170-244,
src/node/delay.rs:990-1019,1076-1113, src/render/graph.rs:708-742."""
import numpy as np
import pytest

import web_audio_api_rs_amd as waa
from graphs import rms_err, white_noise

RQ = 128


def ctx(be, channels, length, sr=48000.0, **kw):
    return waa.OfflineAudioContext(channels, length, sr, binding=be, **kw)


def test_cycle_without_delay_is_muted(be):
    """tests/offline.rs:170-202: gain <-> gain cycle is muted, the other source still renders"""
    c = ctx(be, 1, RQ)
    cycle1 = c.create_gain()
    cycle1.connect(c.destination())
    cycle2 = c.create_gain()
    cycle2.connect(cycle1)
    cycle1.connect(cycle2)
    source_cycle = c.create_constant_source(offset=1.0)
    source_cycle.connect(cycle1)
    other = c.create_constant_source(offset=2.0)
    other.connect(c.destination())
    source_cycle.start()
    other.start()
    out = c.start_rendering_sync().data[0, 0]
    assert np.array_equal(out, np.full(RQ, 2.0, np.float32))


def test_cycle_breaker(be):
    """tests/offline.rs:204-244: delay.connect(&delay), positive feedback: 1, 2, 3 per quantum"""
    sr = 48000.0
    c = ctx(be, 1, RQ * 3, sr)
    delay = c.create_delay(1.0 / sr)
    delay.delay_time.set_value(np.float32(1.0) / np.float32(sr))
    delay.connect(c.destination())
    delay.connect(delay)
    source = c.create_constant_source(offset=1.0)
    source.connect(delay)
    source.connect(c.destination())
    source.start()
    out = c.start_rendering_sync().data[0, 0]
    assert np.array_equal(out[:RQ], np.full(RQ, 1.0, np.float32))
    assert np.array_equal(out[RQ:2 * RQ], np.full(RQ, 2.0, np.float32))
    assert np.array_equal(out[2 * RQ:], np.full(RQ, 3.0, np.float32))


@pytest.mark.parametrize("max_delay_frames,delay_frames", [(48000.0, 1.0), (64.0, 64.0)])
def test_min_delay_when_in_loop(be, max_delay_frames, delay_frames):
    """delay.rs:990-1019 and :1076-1113: inside a loop the delay is clamped to one render quantum, abs_all <= 0"""
    sr = 48000.0
    c = ctx(be, 1, 256, sr)
    delay = c.create_delay(max_delay_frames / sr)
    delay.delay_time.set_value(np.float32(delay_frames) / np.float32(sr))
    delay.connect(c.destination())
    gain = c.create_gain(gain=0.0)  # a loop without feedback
    delay.connect(gain)
    gain.connect(delay)
    src = c.create_buffer_source()
    src.connect(delay)
    src.set_buffer(waa.AudioBuffer(np.array([[1.0]], np.float32), sr))
    src.start_at(0.0)
    out = c.start_rendering_sync().data[0, 0]
    exp = np.zeros(256, np.float32)
    exp[128] = 1.0
    assert np.array_equal(out, exp)


def test_detached_leg_of_a_muted_cycle_still_renders(be):
    """graph.rs:708-742: 4->2, 2->1, 1->0, 1->2, 3->0: nodes 1 and 2 are dropped, 3 renders"""
    c = ctx(be, 1, RQ)
    n1, n2 = c.create_gain(), c.create_gain()
    n3 = c.create_constant_source(offset=0.5)
    n4 = c.create_constant_source(offset=4.0)
    n4.connect(n2)
    n2.connect(n1)
    n1.connect(c.destination())
    n1.connect(n2)
    n3.connect(c.destination())
    n3.start()
    n4.start()
    out = c.start_rendering_sync().data[0, 0]
    assert np.array_equal(out, np.full(RQ, 0.5, np.float32))


def test_feedback_echo_matches_closed_form(be):
    """src -> (+) -> delay(D) -> gain(g) -> back to (+); output = the delay's output: y[n] = x[n-D] + g y[n-D]
    (sample rate 2^15 so that D / sr is exact in f32 and the delay interpolates with k == 0)"""
    sr, n, D, g = 32768.0, RQ * 12, 300, np.float32(0.5)
    x = white_noise(1, 1, n, seed0=4)[0, 0]
    c = ctx(be, 1, n, sr)
    src = c.create_buffer_source()
    src.set_buffer(waa.AudioBuffer(x[None, :], sr))
    delay = c.create_delay(0.1)
    delay.delay_time.set_value(np.float32(D) / np.float32(sr))
    fb = c.create_gain(gain=float(g))
    src.connect(delay)
    delay.connect(fb).connect(delay)
    delay.connect(c.destination())
    src.start()
    out = c.start_rendering_sync().data[0, 0]
    y = np.zeros(n, np.float32)
    for i in range(D, n):
        y[i] = np.float32(x[i - D] + np.float32(g * y[i - D]))
    assert np.max(np.abs(out - y)) <= 1e-6


# --------------------------------------------------------------------------- GPU parity on seeded inputs
def _feedback_graph(binding, noise, delays, gains, with_filter, max_delay=0.05, want_plan=False):
    n = noise.shape[0]
    c = waa.OfflineAudioContext(2, noise.shape[2], 48000.0, n_instances=n, binding=binding)
    src = c.create_buffer_source()
    src.set_buffer_batch(noise, 48000.0)
    delay = c.create_delay(max_delay)
    fb = c.create_gain()
    for i in range(n):
        delay.delay_time.set_value(delays[i], instance=i)
        fb.gain.set_value(gains[i], instance=i)
    src.connect(delay)
    tail = delay
    if with_filter:
        tail = delay.connect(c.create_biquad_filter(type_="lowpass", frequency=3000.0))
    tail.connect(fb).connect(delay)
    src.connect(c.destination())
    tail.connect(c.destination())
    src.start()
    plan = c.plan_describe() if want_plan else None
    out = c.start_rendering_sync().data
    c.close()
    return (out, plan) if want_plan else out


@pytest.mark.gpu
@pytest.mark.parametrize("with_filter", [False, True])
def test_parity_feedback_delay(hip, orc, with_filter):
    n, frames = 6, RQ * 40 + 17
    noise = white_noise(n, 2, frames, seed0=6)
    delays = np.float32([0.0, 0.001, 128.0 / 48000.0, 0.004, 0.0123, 0.05])
    gains = np.float32([0.5, -0.7, 0.9, 0.3, 0.6, 0.8])
    g = _feedback_graph(hip, noise, delays, gains, with_filter)
    o = _feedback_graph(orc, noise, delays, gains, with_filter)
    assert rms_err(g, o).max() <= 1e-6
    assert np.abs(g - o).max() <= (1e-6 if with_filter else 0.0)


def _ping_pong(binding, noise, lfo=None):
    """stereo ping-pong: two DelayNodes in ONE loop (only the first one found is cut, the second keeps its
    writer->reader edge and may use a sub-quantum delay), a WaveShaper and a StereoPanner inside the loop, a
    k-rate swept Biquad, delayTime of the first delay modulated from OUTSIDE the loop."""
    n, _, frames = noise.shape
    nq = (frames + RQ - 1) // RQ
    c = waa.OfflineAudioContext(2, frames, 48000.0, n_instances=n, binding=binding)
    src = c.create_buffer_source()
    src.set_buffer_batch(noise, 48000.0)
    d1 = c.create_delay(0.02, delay_time=0.006)
    d2 = c.create_delay(0.02, delay_time=0.0005)  # 24 frames: sub-quantum, legal for the uncut delay
    ws = c.create_wave_shaper(curve=np.tanh(np.linspace(-2.0, 2.0, 257)).astype(np.float32))
    pan = c.create_stereo_panner(pan=-0.3)
    bq = c.create_biquad_filter(type_="bandpass", frequency=1500.0, q=0.8)
    bq.frequency.set_block(0, np.geomspace(500.0, 4000.0, nq).astype(np.float32))
    fb = c.create_gain(gain=0.6)
    src.connect(d1)
    d1.connect(ws).connect(pan).connect(d2).connect(bq).connect(fb).connect(d1)
    if lfo is not None:
        mod = c.create_buffer_source()
        mod.set_buffer_batch(lfo, 48000.0)
        depth = c.create_gain(gain=0.002)
        mod.connect(depth).connect(d1.delay_time)
        mod.start()
    src.connect(c.destination())
    d2.connect(c.destination())
    src.start()
    plan = c.plan_describe() if binding.prefix == "waa_" else ""
    out = c.start_rendering_sync().data
    c.close()
    return out, plan


@pytest.mark.gpu
@pytest.mark.parametrize("modulated", [False, True])
def test_parity_ping_pong_loop(hip, orc, modulated):
    n, frames = 3, RQ * 48
    noise = (white_noise(n, 2, frames, seed0=12) * 0.5).astype(np.float32)
    lfo = None
    if modulated:
        t = np.arange(frames) / 48000.0
        lfo = np.stack([np.sin(2 * np.pi * (2.0 + i) * t) for i in range(n)]).astype(np.float32)[:, None, :]
    g, plan = _ping_pong(hip, noise, lfo)
    o, _ = _ping_pong(orc, noise, lfo)
    assert "feedback loop: 8 item(s)" in plan and "(clamped)" in plan
    assert np.isfinite(o).all()
    assert rms_err(g, o).max() <= 1e-6
    assert np.abs(g - o).max() <= 5e-6


@pytest.mark.gpu
def test_node_kinds_outside_the_static_loop_kernel_go_to_the_dynamic_path(hip, orc):
    """An IIRFilter inside a short feedback loop: the static loop kernel does not cover it (status 4 in round 1); the
    planner now renders the whole graph quantum by quantum with dyn_kernel.  A ConvolverNode inside a loop whose delay is
    shorter than one partition of its impulse response was refused until round 4; since round 5 a response with 128-frame
    partitions follows the loop quantum by quantum (tests/test_frozen_loops.py) — a LONGER response (partitions of several
    quanta) in such a loop is still refused."""
    outs = []
    for be in (hip, orc):
        c = waa.OfflineAudioContext(2, RQ * 40, 48000.0, n_instances=2, binding=be)
        src = c.create_constant_source()
        d = c.create_delay(0.1, delay_time=0.01)
        iir = c.create_iir_filter([0.5, 0.5], [1.0, -0.2])
        src.connect(d)
        d.connect(iir).connect(d)
        d.connect(c.destination())
        src.start()
        if be is hip:
            assert "dynamic-count group" in c.plan_describe()
        outs.append(c.start_rendering_sync().data)
        c.close()
    assert np.abs(outs[0] - outs[1]).max() <= 1e-6
    c = waa.OfflineAudioContext(2, RQ * 4, 48000.0, n_instances=1, binding=hip)
    src = c.create_constant_source()
    d = c.create_delay(0.1, delay_time=0.01)
    conv = c.create_convolver(buffer=waa.AudioBuffer(np.ones((1, 5000), np.float32), 48000.0))
    src.connect(d)
    d.connect(conv).connect(d)
    d.connect(c.destination())
    src.start()
    with pytest.raises(waa.WaaError) as ei:
        c.start_rendering_sync()
    assert ei.value.status == 4


def _convolver_loop(binding, noise, ir, delay_s, fb_gain, frames, with_filter=False, device=-1):
    """src -> Delay -> Convolver -> [Biquad] -> Gain -> back into the Delay; the destination hears the convolver and the
    dry delay (a reverb inside an echo)"""
    n, n_ch, _ = noise.shape
    c = waa.OfflineAudioContext(2, frames, 48000.0, n_instances=n, binding=binding, device=device)
    src = c.create_buffer_source()
    src.set_buffer_batch(noise, 48000.0)
    d = c.create_delay(1.0, delay_time=delay_s)
    conv = c.create_convolver(buffer=waa.AudioBuffer(ir, 48000.0), disable_normalization=True)
    fb = c.create_gain(gain=fb_gain)
    src.connect(d)
    tail = d.connect(conv)
    if with_filter:
        tail = tail.connect(c.create_biquad_filter(type_="highpass", frequency=300.0))
    tail.connect(fb).connect(d)
    tail.connect(c.destination())
    d.connect(c.destination())
    src.start()
    return c


def _decaying_ir(n_ch, taps, seed, level):
    rng = np.random.default_rng(seed)
    env = np.exp(-4.0 * np.arange(taps) / taps)
    return (rng.standard_normal((n_ch, taps)) * env * level).astype(np.float32)


CONV_LOOPS = {
    # name: (IR channels, IR taps, source channels, delay [s], feedback gain, with a Biquad in the loop)
    "direct-fir-64": (1, 64, 2, 0.05, 0.6, False),            # 2400-frame delay: 1 tile per block, direct FIR pieces
    "fft-128x24": (2, 3000, 2, 0.1, 0.5, True),               # B=128: 2 tiles per block = 32 partitions per launch
    "fft-512-mono": (1, 9000, 1, 0.15, 0.5, False),           # B=512, mono all the way round (mono into a stereo IR would
                                                              # change the loop's channel count after the first echo: dynamic)
    "fft-2048": (2, 40000, 2, 0.2, 0.4, False),               # B=2048: 4 tiles per block
    "fft3-8192-true-stereo": (4, 60000, 2, 0.5, 0.5, True),   # B=8192 (the three-pass transforms): 8 of 11 tiles per block
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CONV_LOOPS))
def test_parity_convolver_inside_a_block_scheduled_loop(hip, orc, name):
    """a ConvolverNode inside a feedback loop whose delay is longer than one partition: the forward transforms, the
    partition sums and the inverse transforms run per block range of the loop (the spectra of earlier blocks stay in X)"""
    ir_ch, taps, src_ch, delay_s, fb_gain, with_filter = CONV_LOOPS[name]
    n, frames = 3, 2048 * 26 + 777
    noise = white_noise(n, src_ch, frames, seed0=91)   # (a source that ends makes the channel counts dynamic: refused, below)
    ir = _decaying_ir(ir_ch, taps, 5, 2.0 / np.sqrt(taps))
    outs = []
    for be in (hip, orc):
        c = _convolver_loop(be, noise, ir, delay_s, fb_gain, frames, with_filter)
        if be is hip:
            plan = c.plan_describe()
            assert "block-scheduled" in plan and "convolver node" in plan and "dynamic-count" not in plan, plan
        outs.append(c.start_rendering_sync().data)
        c.close()
    g, o = outs
    assert np.isfinite(o).all() and np.abs(o).max() > 0.1
    scale = max(1.0, float(np.abs(o).max()))
    assert rms_err(g, o).max() / scale <= 1e-6
    assert np.abs(g - o).max() / scale <= 2e-5


def test_plan_convolver_inside_a_loop(hip):
    """plan only (CPU): the block is a whole number of partitions; a loop delay shorter than a partition of a LONG response is refused"""
    noise = white_noise(2, 2, 2048 * 30)
    c = _convolver_loop(hip, noise, _decaying_ir(2, 60000, 1, 0.01), 0.5, 0.5, 2048 * 30, device=waa.PLAN_ONLY)
    plan = c.plan_describe()
    assert "fft B=8192" in plan and "block-scheduled, 8 tile(s)" in plan, plan   # 24000-frame delay: 11 tiles, rounded to 2 partitions
    c.close()
    c = _convolver_loop(hip, noise, _decaying_ir(2, 3000, 1, 0.01), 0.1, 0.5, 2048 * 30, device=waa.PLAN_ONLY)
    plan = c.plan_describe()
    assert "fft B=128" in plan and "block-scheduled, 2 tile(s)" in plan, plan
    c.close()
    c = _convolver_loop(hip, noise, _decaying_ir(2, 60000, 1, 0.01), 0.1, 0.5, 2048 * 30, device=waa.PLAN_ONLY)   # 4800 < 8192
    with pytest.raises(waa.WaaError) as ei:
        c.plan_describe()
    assert ei.value.status == 4 and "ConvolverNode inside a feedback loop" in str(ei.value)
    c.close()
    # channel counts that change during the render (the source ends) need the quantum-serial dynamic path: since round 5 a
    # response with 128-frame partitions follows the loop quantum by quantum there (tests/test_frozen_loops.py) ...
    c = _convolver_loop(hip, noise[:, :, :2048 * 8], _decaying_ir(2, 3000, 1, 0.01), 0.1, 0.5, 2048 * 30, device=waa.PLAN_ONLY)
    plan = c.plan_describe()
    assert "fft B=128" in plan and "cut at the node(s)" in plan and "render quanta (the shortest delay across a cut" in plan, plan
    c.close()
    # ... a longer response (partitions of several quanta) is still refused
    c = _convolver_loop(hip, noise[:, :, :2048 * 8], _decaying_ir(2, 9000, 1, 0.01), 0.1, 0.5, 2048 * 30, device=waa.PLAN_ONLY)
    with pytest.raises(waa.WaaError) as ei:
        c.plan_describe()
    assert ei.value.status == 4 and "ConvolverNode inside a feedback loop" in str(ei.value)
    c.close()


@pytest.mark.measure
@pytest.mark.gpu
def test_parity_echo_loop_as_one_persistent_launch(hip, orc, monkeypatch):
    """WAA_PERSISTENT_LOOP=1: the single-launch-per-block echo loop walks its blocks inside ONE launch (one workgroup per
    instance, a barrier between blocks); bit-identical to the launch-per-block form"""
    n, frames = 5, 2048 * 9 + 300
    noise = white_noise(n, 2, frames, seed0=33)
    delays = np.float32([0.0430, 0.05, 0.0861, 0.1, 0.0999])
    gains = np.float32([0.5, -0.7, 0.9, 0.3, 0.6])
    plain = _feedback_graph(hip, noise, delays, gains, False)
    monkeypatch.setenv("WAA_PERSISTENT_LOOP", "1")
    pers = _feedback_graph(hip, noise, delays, gains, False)
    assert np.array_equal(plain, pers)
    o = _feedback_graph(orc, noise, delays, gains, False)
    assert np.abs(pers - o).max() == 0.0


def _echo_graph(binding, noise, delays, gains, variant, out_channels=2):
    """source -> Delay <-> Gain, the destination fed as `variant` says:
    dry+wet      source and delay straight into the destination (the tail the ring kernel can render itself)
    wet-only     delay -> destination only
    wet-gain     delay -> Gain(0.7) -> destination (the reader has an op of its own: it stays a launch, the line is stored)
    two-readers  delay -> destination and delay -> Biquad -> destination (the line has two readers: it must be stored)
    other-dry    delay -> destination plus a SECOND source -> destination (a tail input the loop does not read)
    two-sources  a second source feeds the delay too; the destination gets that second source + delay"""
    n = noise.shape[0]
    c = waa.OfflineAudioContext(out_channels, noise.shape[2], 48000.0, n_instances=n, binding=binding)
    src = c.create_buffer_source()
    src.set_buffer_batch(noise, 48000.0)
    delay = c.create_delay(0.4)
    fb = c.create_gain()
    for i in range(n):
        delay.delay_time.set_value(delays[i], instance=i)
        fb.gain.set_value(gains[i], instance=i)
    src.connect(delay)
    delay.connect(fb).connect(delay)
    if variant == "dry+wet":
        src.connect(c.destination())
        delay.connect(c.destination())
    elif variant == "wet-only":
        delay.connect(c.destination())
    elif variant == "wet-gain":
        delay.connect(c.create_gain(gain=0.7)).connect(c.destination())
    elif variant == "two-readers":
        delay.connect(c.destination())
        delay.connect(c.create_biquad_filter(type_="lowpass", frequency=3000.0)).connect(c.destination())
    elif variant == "other-dry":
        other = c.create_buffer_source()
        other.set_buffer_batch(noise[:, :, ::-1].copy(), 48000.0)
        other.connect(c.destination())
        other.start()
        delay.connect(c.destination())
    elif variant == "two-sources":
        other = c.create_buffer_source()
        other.set_buffer_batch(noise[:, :, ::-1].copy(), 48000.0)
        other.connect(delay)
        other.connect(c.destination())
        other.start()
        delay.connect(c.destination())
    src.start()
    plan = c.plan_describe() if binding.prefix == "waa_" else ""
    out = c.start_rendering_sync().data
    c.close()
    return out, plan


@pytest.mark.measure
@pytest.mark.gpu
@pytest.mark.parametrize("channels,out_channels", [(1, 1), (1, 2), (2, 2)])
@pytest.mark.parametrize("variant", ["dry+wet", "wet-only", "wet-gain", "two-readers", "other-dry", "two-sources"])
def test_parity_echo_loop_from_the_lds_ring(hip, orc, channels, out_channels, variant, monkeypatch):
    """an echo loop whose body is ONE element-wise step and whose delays all fit the 16384-frame window is rendered by
    waa_echo.hip in one launch, the delayed read served from LDS; when the line's only other reader is a sum of the
    delayed line and signals the loop reads (dry + wet into the destination) the same launch renders that too and the
    line is never stored.  Bit-identical to the launch-per-block form and to the oracle, at the window's limits (2064
    frames = the shortest block-scheduled delay, 14328 = 16384-8*256-8 frames with 2048-frame chunks), ragged tail"""
    n, frames = 6, 2048 * 11 + 77
    noise = white_noise(n, channels, frames, seed0=35)
    delays = (np.float64([2064, 2065.5, 3000.25, 4800, 9000.75, 14328]) / 48000.0).astype(np.float32)
    gains = np.float32([0.5, -0.7, 0.9, 0.3, 0.6, -0.95])
    ring, plan = _echo_graph(hip, noise, delays, gains, variant, out_channels)
    assert "LDS-ring kernel in ONE launch" in plan and "chunks of 2048 frames" in plan
    assert ("the line is not stored" in plan) == (variant in ("dry+wet", "wet-only", "two-sources"))
    o, _ = _echo_graph(orc, noise, delays, gains, variant, out_channels)
    tol = 2e-6 if variant == "two-readers" else 0.0   # (the Biquad branch: f64 recurrence, other summation order)
    assert np.abs(ring - o).max() <= tol
    monkeypatch.setenv("WAA_NO_ECHO_TAIL", "1")
    untailed, plan = _echo_graph(hip, noise, delays, gains, variant, out_channels)
    assert "LDS-ring kernel in ONE launch" in plan and "the line is not stored" not in plan
    assert np.array_equal(untailed, ring)
    monkeypatch.setenv("WAA_NO_ECHO_RING", "1")
    plain, plan = _echo_graph(hip, noise, delays, gains, variant, out_channels)
    assert "LDS-ring kernel" not in plan
    assert np.array_equal(plain, ring)


def _random_echo_graph(binding, seed, plan_only=False, short=False, filtered=False):
    """a random member of the echo-loop family: 1-2 sources (mono / stereo, optionally through a Gain that rides on the
    delay's input edge) into Delay <-> Gain, connection order (= summation order) shuffled, feedback gain constant / per
    instance / one value per quantum (one of them sometimes 0 or 1: gain.rs' mute and pass-through cases), speakers or
    discrete up-mix, an explicit stereo line over a mono source, and one of several destinations"""
    rng = np.random.default_rng(seed)
    n = 3
    frames = 2048 * int(rng.integers(6, 11)) + int(rng.integers(0, 300))
    nq = (frames + RQ - 1) // RQ
    out_ch = int(rng.integers(1, 3))
    c = waa.OfflineAudioContext(out_ch, frames, 48000.0, n_instances=n, binding=binding, **({"device": waa.PLAN_ONLY} if plan_only else {}))
    n_src = int(rng.integers(1, 3))
    delay = c.create_delay(0.4)
    if rng.random() < 0.3:
        delay.set_channel_interpretation("discrete")
    if rng.random() < 0.25:
        delay.set_channel_count(2)
        delay.set_channel_count_mode("explicit")
    for i in range(n):
        # (short: below one 2048-frame tile — comb filters, plucked strings: the ring kernel in 256- / 512-frame chunks)
        delay.delay_time.set_value(np.float32((rng.uniform(137.0, 2040.0) if short else rng.uniform(2057.0, 14000.0)) / 48000.0), instance=i)
    fb = c.create_gain()
    bq = None
    if filtered:  # Delay -> Biquad -> Gain -> back (the ring kernel's BQ form)
        bq = c.create_biquad_filter(type_=str(rng.choice(["lowpass", "highpass", "bandpass", "peaking", "allpass"])),
                                    frequency=float(np.exp(rng.uniform(np.log(40.0), np.log(9000.0)))))   # (down to filters whose memory outlasts many chunks)
        bq.q.set_value(float(rng.uniform(0.3, 4.0)))
        bq.gain.set_value(float(rng.uniform(-6.0, 6.0)))
    kind = rng.integers(0, 3)
    if kind == 0:
        fb.gain.set_value(float(rng.choice([0.5, -0.8, 0.0, 1.0])))
    elif kind == 1:
        for i in range(n):
            fb.gain.set_value(np.float32(rng.uniform(-0.95, 0.95)), instance=i)
    else:
        g = rng.uniform(-0.9, 0.9, nq).astype(np.float32)
        g[rng.integers(0, nq, 6)] = rng.choice([0.0, 1.0], 6)
        fb.gain.set_block(0, g)
    connects = []
    srcs = []
    for k in range(n_src):
        ch = int(rng.integers(1, 3))
        src = c.create_buffer_source()
        src.set_buffer_batch(white_noise(n, ch, frames, seed0=seed * 7 + k), 48000.0)
        src.start()
        srcs.append(src)
        head = src
        if rng.random() < 0.4:
            head = src.connect(c.create_gain(gain=float(rng.choice([0.7, -1.3, 1.0]))))
        connects.append(lambda head=head: head.connect(delay))
    connects.append(lambda: (delay.connect(bq).connect(fb) if bq is not None else delay.connect(fb)).connect(delay))
    wet = bq if bq is not None else delay   # what the destinations below listen to
    for k in rng.permutation(len(connects)):
        connects[k]()
    dest = int(rng.integers(0, 5))
    if dest == 0:      # dry + wet
        srcs[0].connect(c.destination())
        wet.connect(c.destination())
    elif dest == 1:    # wet + dry, the other order
        wet.connect(c.destination())
        srcs[-1].connect(c.destination())
    elif dest == 2:    # wet only
        wet.connect(c.destination())
    elif dest == 3:    # a reader with an op of its own
        wet.connect(c.create_gain(gain=0.6)).connect(c.destination())
    else:              # two readers
        wet.connect(c.destination())
        delay.connect(c.create_wave_shaper(curve=np.float32([-0.5, 0.0, 0.8]))).connect(c.destination())
    plan = c.plan_describe() if binding.prefix == "waa_" else ""
    out = None if plan_only else c.start_rendering_sync().data
    c.close()
    return out, plan


def test_random_echo_loops_mostly_qualify_for_the_ring(hip):
    """(CPU) the family of the GPU test below is planned as intended: most members through the LDS-ring kernel, a good part
    with the tail stage, nothing refused"""
    plans = [_random_echo_graph(hip, 9000 + seed, plan_only=True)[1] for seed in range(40)]
    ring = sum("LDS-ring kernel in ONE launch" in p for p in plans)
    fused = sum("the line is not stored" in p for p in plans)
    assert ring >= 25 and fused >= 8, (ring, fused)


@pytest.mark.gpu
def test_parity_random_echo_loops(hip, orc):
    """40 random members of the echo-loop family, every one bit-identical to the oracle; most of them through the LDS-ring
    kernel, a good part with the tail stage"""
    ring = fused = 0
    for seed in range(40):
        g, plan = _random_echo_graph(hip, 9000 + seed)
        o, _ = _random_echo_graph(orc, 9000 + seed)
        assert np.array_equal(g, o), (seed, float(np.abs(g - o).max()), plan)
        ring += "LDS-ring kernel in ONE launch" in plan
        fused += "the line is not stored" in plan
    assert ring >= 25 and fused >= 8, (ring, fused)


@pytest.mark.gpu
@pytest.mark.parametrize("short,filtered", [(True, False), (False, True), (True, True)])
def test_parity_random_echo_loops_short_and_filtered(hip, orc, short, filtered):
    """the same family with delays below a tile (137 .. 2040 frames: chunks of 128 / 256 / 512 frames, rings of 1024 .. 4096) and / or a Biquad
    in the loop (the BQ form): plain loops bit-identical to the oracle, filtered ones within the streaming Biquad's tolerance"""
    ring = 0
    for seed in range(40):
        g, plan = _random_echo_graph(hip, 9500 + seed, short=short, filtered=filtered)
        o, _ = _random_echo_graph(orc, 9500 + seed, short=short, filtered=filtered)
        if filtered:
            scale = max(1.0, float(np.abs(o).max()))
            assert rms_err(g, o).max() <= 1e-6 * scale and np.abs(g - o).max() <= 4e-6 * scale, (seed, float(np.abs(g - o).max()), plan)
        else:
            assert np.array_equal(g, o), (seed, float(np.abs(g - o).max()), plan)
        ring += "LDS-ring kernel in ONE launch" in plan
        assert ("shorter than a tile" in plan) == (short and "LDS-ring kernel in ONE launch" in plan), plan
    assert ring >= 15, ring


@pytest.mark.gpu
def test_echo_loop_past_the_lds_ring_window(hip, orc):
    """one instance's delay is past what the smallest chunk reaches (16384 - 128 - 8 = 16248 frames): the
    launch-per-block form renders the loop"""
    n, frames = 6, 2048 * 11 + 77
    noise = white_noise(n, 2, frames, seed0=36)
    delays = (np.float64([2064, 2065.5, 3000.25, 4800, 9000.75, 16250]) / 48000.0).astype(np.float32)
    gains = np.float32([0.5, -0.7, 0.9, 0.3, 0.6, -0.95])
    g, plan = _echo_graph(hip, noise, delays, gains, "dry+wet")
    assert "LDS-ring kernel" not in plan
    assert np.abs(g - _echo_graph(orc, noise, delays, gains, "dry+wet")[0]).max() == 0.0


@pytest.mark.measure
@pytest.mark.gpu
@pytest.mark.parametrize("forced_loop_kernel", [False, True])
@pytest.mark.parametrize("with_filter", [False, True])
def test_parity_long_feedback_delay_block_scheduled(hip, orc, with_filter, forced_loop_kernel, monkeypatch):
    """every loop delay is longer than a 2048-frame tile: the loop is rendered block by block with the ordinary
    node-major kernels (streaming biquad included); WAA_LOOP_KERNEL forces the quantum-serial kernel instead"""
    if forced_loop_kernel:
        monkeypatch.setenv("WAA_LOOP_KERNEL", "1")
    n, frames = 5, 2048 * 7 + 300
    noise = white_noise(n, 2, frames, seed0=31)
    delays = np.float32([0.0430, 0.05, 0.0861, 0.1, 0.0999])   # 2064 .. 4800 frames
    gains = np.float32([0.5, -0.7, 0.9, 0.3, 0.6])
    g = _feedback_graph(hip, noise, delays, gains, with_filter)
    o = _feedback_graph(orc, noise, delays, gains, with_filter)
    assert rms_err(g, o).max() <= 1e-6
    assert np.abs(g - o).max() <= (2e-6 if with_filter else 0.0)


def test_plan_block_scheduled_loop(hip):
    c = waa.OfflineAudioContext(2, 2048 * 8, 48000.0, n_instances=2, binding=hip, device=waa.PLAN_ONLY)
    src = c.create_buffer_source()
    src.set_buffer_batch(white_noise(2, 2, 2048 * 8), 48000.0)
    delay = c.create_delay(1.0, delay_time=0.25)          # 12000 frames -> blocks of 5 tiles
    bq = c.create_biquad_filter(type_="lowpass", frequency=3000.0)
    fb = c.create_gain(gain=0.5)
    src.connect(delay)
    delay.connect(bq).connect(fb).connect(delay)
    bq.connect(c.destination())
    src.start()
    plan = c.plan_describe()
    assert "block-scheduled, 5 tile(s) = 10240 frames per block, 3 step(s) per block" in plan
    # the loop-breaking delay is read from its line by its consumer, the gain rides on an input edge of the delay's mix
    assert "biquad_stream" in plan and "inside a block-scheduled loop" in plan and "gain node" in plan and "delayed:2ch" in plan
    c.close()


@pytest.mark.measure
@pytest.mark.parametrize("variant,fused", [("dry+wet", True), ("wet-only", True), ("wet-gain", False), ("two-readers", False),
                                            ("other-dry", False)])
def test_plan_echo_loop_ring_and_tail(hip, variant, fused, monkeypatch):
    """which echo loops the planner hands to the LDS-ring kernel (waa_echo.hip), and which readers of the line it renders in
    the same launch — decided on the launch list, so a plan-only context shows it"""
    def plan_of(delay_frames):
        c = waa.OfflineAudioContext(2, 2048 * 8, 48000.0, n_instances=2, binding=hip, device=waa.PLAN_ONLY)
        src = c.create_buffer_source()
        src.set_buffer_batch(white_noise(2, 2, 2048 * 8), 48000.0)
        delay = c.create_delay(1.0, delay_time=delay_frames / 48000.0)
        src.connect(delay)
        delay.connect(c.create_gain(gain=0.5)).connect(delay)
        if variant == "dry+wet":
            src.connect(c.destination())
        if variant == "wet-gain":
            delay.connect(c.create_gain(gain=0.7)).connect(c.destination())
        else:
            delay.connect(c.destination())
        if variant == "two-readers":
            delay.connect(c.create_biquad_filter(type_="lowpass", frequency=3000.0)).connect(c.destination())
        if variant == "other-dry":
            other = c.create_buffer_source()
            other.set_buffer_batch(white_noise(2, 2, 2048 * 8, seed0=3), 48000.0)
            other.connect(c.destination())
            other.start()
        src.start()
        plan = c.plan_describe()
        c.close()
        return plan
    plan = plan_of(12000)
    assert "LDS-ring kernel in ONE launch: delay 12000 .. 12000 frames, chunks of 4096 frames" in plan
    assert ("the line is not stored" in plan) == fused
    if variant == "two-readers":
        assert "the delay line has 2 reader(s) outside the loop" in plan
    if variant in ("wet-gain", "other-dry"):
        assert "is not a plain sum of the delayed line and of the loop's inputs" in plan
    assert "chunks of 2048 frames" in plan_of(2064) and "chunks of 512 frames" in plan_of(15354) and "LDS-ring" not in plan_of(16250)
    monkeypatch.setenv("WAA_NO_ECHO_TAIL", "1")
    assert "LDS-ring kernel in ONE launch" in plan_of(12000) and "the line is not stored" not in plan_of(12000)
    monkeypatch.setenv("WAA_NO_ECHO_RING", "1")
    assert "LDS-ring" not in plan_of(12000)


@pytest.mark.measure
def test_plan_block_scheduled_loop_node_major_delay(hip, monkeypatch):
    """the switches: every loop member a launch of its own (the round-1 form)"""
    monkeypatch.setenv("WAA_NO_LOOP_FOLD", "1")
    c = waa.OfflineAudioContext(2, 2048 * 8, 48000.0, n_instances=2, binding=hip, device=waa.PLAN_ONLY)
    src = c.create_buffer_source()
    src.set_buffer_batch(white_noise(2, 2, 2048 * 8), 48000.0)
    delay = c.create_delay(1.0, delay_time=0.25)
    bq = c.create_biquad_filter(type_="lowpass", frequency=3000.0)
    fb = c.create_gain(gain=0.5)
    src.connect(delay)
    delay.connect(bq).connect(fb).connect(delay)
    bq.connect(c.destination())
    src.start()
    plan = c.plan_describe()
    assert "block-scheduled, 5 tile(s) = 10240 frames per block, 4 step(s) per block" in plan
    assert "biquad_stream" in plan and "in a loop: clamped" in plan
    c.close()


def _self_modulated_loop(binding, noise, delay_s, device=-1):
    """the loop's own signal modulates its feedback gain: src -> Delay -> Gain(g) -> back, Delay -> Gain(depth) -> g.gain"""
    n, _, frames = noise.shape
    c = waa.OfflineAudioContext(2, frames, 48000.0, n_instances=n, binding=binding, device=device)
    src = c.create_buffer_source()
    src.set_buffer_batch(noise, 48000.0)
    d = c.create_delay(1.0, delay_time=delay_s)
    g = c.create_gain(gain=0.3)
    depth = c.create_gain(gain=0.2)
    src.connect(d)
    d.connect(g).connect(d)
    d.connect(depth).connect(g.gain)
    d.connect(c.destination())
    src.start()
    return c


@pytest.mark.gpu
def test_parity_param_modulated_from_inside_a_block_scheduled_loop(hip, orc):
    """the param's summing chain reads a signal of the loop: it is a launch of every block, in its place in the order (it was
    classed with the loop's prologue — run once, before the blocks, on data not yet rendered — until round 3)"""
    noise = white_noise(3, 2, 2048 * 9 + 500, seed0=12)
    outs = []
    for be in (hip, orc):
        c = _self_modulated_loop(be, noise, 0.1)
        if be is hip:
            plan = c.plan_describe()
            assert "block-scheduled" in plan and "PARAM_ADD" in plan, plan
        outs.append(c.start_rendering_sync().data)
        c.close()
    g, o = outs
    assert np.abs(o).max() > 0.1
    assert rms_err(g, o).max() <= 1e-6 and np.abs(g - o).max() <= 2e-6


def test_param_modulated_from_inside_a_short_loop(hip):
    """a GainNode's gain driven from inside its own quantum-serial loop is rendered by the dynamic-count kernel since round 6 (parity:
    tests/test_param_modulation.py); a filter's or a DelayNode's param driven that way stays status 4"""
    noise = white_noise(2, 2, 2048 * 4)
    c = _self_modulated_loop(hip, noise, 0.01, device=waa.PLAN_ONLY)   # 480 frames: quantum-serial loop
    plan = c.plan_describe()
    assert "dyn_kernel" in plan and "GAIN" in plan
    c.close()
    c = waa.OfflineAudioContext(2, 2048 * 4, 48000.0, n_instances=2, binding=hip, device=waa.PLAN_ONLY)
    src = c.create_buffer_source()
    src.set_buffer_batch(noise, 48000.0)
    d = c.create_delay(1.0, delay_time=0.01)
    bq = c.create_biquad_filter(type_="lowpass", frequency=800.0)
    src.connect(d)
    d.connect(bq).connect(c.create_gain(gain=0.3)).connect(d)
    d.connect(c.create_gain(gain=200.0)).connect(bq.frequency)
    d.connect(c.destination())
    src.start()
    with pytest.raises(waa.WaaError) as ei:
        c.plan_describe()
    assert ei.value.status == 4 and "modulated from inside its own feedback loop" in str(ei.value)
    c.close()


def _echo_loop_with_an_analyser_on_the_line(binding, noise, device=-1):
    """source -> WaveShaper(no curve: an identity member of the loop) -> Delay -> Gain -> back into the WaveShaper; an
    AnalyserNode on the WaveShaper ALIASES the loop's delay line and is pulled by analyser_kernel, outside the launch
    list; dry + wet into the destination is the line's only launch-side reader (fuse_echo_tails)"""
    n, _, frames = noise.shape
    c = waa.OfflineAudioContext(2, frames, 48000.0, n_instances=n, binding=binding, device=device)
    src = c.create_buffer_source()
    src.set_buffer_batch(noise, 48000.0)
    ws = c.create_wave_shaper()
    delay = c.create_delay(1.0, delay_time=12000 / 48000.0)
    an = c.create_analyser()
    src.connect(ws).connect(delay)
    delay.connect(c.create_gain(gain=0.5)).connect(ws)
    ws.connect(an)
    src.connect(c.destination())
    delay.connect(c.destination())
    src.start()
    return c, an


def test_plan_echo_line_stays_stored_for_an_analyser_that_aliases_it(hip):
    c, _ = _echo_loop_with_an_analyser_on_the_line(hip, white_noise(2, 2, 2048 * 8), device=waa.PLAN_ONLY)
    plan = c.plan_describe()
    c.close()
    assert "LDS-ring kernel in ONE launch" in plan and "alias node" in plan
    assert "aliases the loop's delay line" in plan and "the line is not stored" not in plan, plan


def _check_analyser_on_the_echo_line():
    """body of the next test (run in a subprocess on poisoned device memory)"""
    import ctypes
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    orc = waa.bind(ctypes.CDLL(os.path.join(root, "oracle", "liboracle.so")), "orc_")
    noise = white_noise(3, 2, 2048 * 9 + 300, seed0=17)
    outs, bins, times = [], [], []
    for be in (waa.default_binding(), orc):
        c, an = _echo_loop_with_an_analyser_on_the_line(be, noise)
        outs.append(c.start_rendering_sync().data)
        bins.append(an.get_float_frequency_data(instance=2))
        times.append(an.get_float_time_domain_data(instance=2))
        c.close()
    assert np.array_equal(outs[0], outs[1]) and np.abs(outs[1]).max() > 0.1
    assert np.array_equal(times[0], times[1]) and np.abs(times[1]).max() > 0.1
    assert np.abs(bins[0] - bins[1]).max() <= 1e-3
    print("ok")


@pytest.mark.gpu
def test_parity_analyser_that_aliases_the_echo_line_on_poisoned_memory():
    """the analyser's pull reads the line after the render: with WAA_POISON_ALLOC=1 (read once per process, hence the
    subprocess) an unwritten line would be all NaN"""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, WAA_POISON_ALLOC="1", PYTHONPATH=os.pathsep.join([os.path.dirname(here), here]))
    out = subprocess.check_output([sys.executable, "-c", "import test_cycles as t; t._check_analyser_on_the_echo_line()"],
                                  env=env, cwd=here)
    assert out.strip().endswith(b"ok")


def _filtered_echo_graph(binding, noise, delays, gains, variant, out_channels=2, plan_only=False):
    """source -> Delay -> Biquad -> Gain -> back into the Delay (the classic filtered echo), the destination fed as `variant` says:
    dry+wet       the filter's output and the source (the tail the ring kernel renders itself: nothing but the output is stored)
    wet-only      the filter's output only
    wet-gain      filter -> Gain(0.7) -> destination (a reader with an op of its own: the filter's output is stored for it)
    line-reader   the DELAY's output reaches the destination too (read from the delay line, which is stored for that reader)
    two-readers   the filter's output into the destination and into a WaveShaper -> destination (stored)
    two-sources   a second source feeds the delay too (three inputs of the sum)
    peaking       dry+wet with a peaking filter, per-instance frequency and Q"""
    n = noise.shape[0]
    kw = {"device": waa.PLAN_ONLY} if plan_only else {}
    c = waa.OfflineAudioContext(out_channels, noise.shape[2], 48000.0, n_instances=n, binding=binding, **kw)
    src = c.create_buffer_source()
    src.set_buffer_batch(noise, 48000.0)
    delay = c.create_delay(0.4)
    bq = c.create_biquad_filter(type_="peaking" if variant == "peaking" else "lowpass", frequency=2500.0)
    fb = c.create_gain()
    for i in range(n):
        delay.delay_time.set_value(delays[i], instance=i)
        fb.gain.set_value(gains[i], instance=i)
        if variant == "peaking":
            bq.frequency.set_value(700.0 + 900.0 * i, instance=i)
            bq.q.set_value(0.7 + 0.4 * i, instance=i)
            bq.gain.set_value(4.0 - 2.0 * i, instance=i)
    src.connect(delay)
    delay.connect(bq).connect(fb).connect(delay)
    if variant in ("dry+wet", "peaking"):
        bq.connect(c.destination())
        src.connect(c.destination())
    elif variant == "wet-only":
        bq.connect(c.destination())
    elif variant == "wet-gain":
        bq.connect(c.create_gain(gain=0.7)).connect(c.destination())
    elif variant == "line-reader":
        bq.connect(c.destination())
        delay.connect(c.destination())
    elif variant == "two-readers":
        bq.connect(c.destination())
        bq.connect(c.create_wave_shaper(curve=np.float32([-0.5, 0.0, 0.8]))).connect(c.destination())
    elif variant == "two-sources":
        other = c.create_buffer_source()
        other.set_buffer_batch(noise[:, :, ::-1].copy(), 48000.0)
        other.connect(delay)
        other.connect(c.destination())
        other.start()
        bq.connect(c.destination())
    src.start()
    plan = c.plan_describe() if binding.prefix == "waa_" else ""
    out = None if plan_only else c.start_rendering_sync().data
    c.close()
    return out, plan


FILTERED_ECHOES = ["dry+wet", "wet-only", "wet-gain", "line-reader", "two-readers", "two-sources", "peaking"]


@pytest.mark.measure
@pytest.mark.parametrize("variant", FILTERED_ECHOES)
def test_plan_filtered_echo_loop(variant):
    """which filtered echo loops become ONE launch of the ring kernel's BQ form, and what leaves that launch"""
    n, frames = 6, 2048 * 11 + 77
    noise = white_noise(n, 2, frames, seed0=37)
    delays = (np.float64([2064, 2065.5, 3000.25, 4800, 9000.75, 14328]) / 48000.0).astype(np.float32)
    gains = np.float32([0.5, -0.7, 0.9, 0.3, 0.6, -0.95])
    _, plan = _filtered_echo_graph(waa.measure_binding(), noise, delays, gains, variant, plan_only=True)
    assert "with the Biquad between the delayed read and the sum" in plan, plan
    assert ("the delay line is not stored" in plan) == (variant != "line-reader"), plan
    # (wet-only: the destination IS the filter's output — stored, as the render's result)
    fused = variant in ("dry+wet", "two-sources", "peaking")
    assert ("the filter's output is not stored" in plan) == fused, plan


@pytest.mark.measure
@pytest.mark.gpu
@pytest.mark.parametrize("channels,out_channels", [(1, 1), (1, 2), (2, 2)])
@pytest.mark.parametrize("variant", FILTERED_ECHOES)
def test_parity_filtered_echo_loop_from_the_lds_ring(hip, orc, channels, out_channels, variant, monkeypatch):
    """waa_echo.hip's BQ form against the oracle and against the three launches per block it replaces (WAA_NO_ECHO_BQ): the
    Biquad's f64 recurrence is evaluated in the reference's order from incoming states found by a scan, like the streaming
    kernel — not bit-identical to the serial recurrence, hence the streaming kernel's tolerance"""
    n, frames = 6, 2048 * 11 + 77
    noise = white_noise(n, channels, frames, seed0=38)
    delays = (np.float64([2064, 2065.5, 3000.25, 4800, 9000.75, 14328]) / 48000.0).astype(np.float32)
    gains = np.float32([0.5, -0.7, 0.9, 0.3, 0.6, -0.95])
    ring, plan = _filtered_echo_graph(hip, noise, delays, gains, variant, out_channels)
    assert "with the Biquad between the delayed read and the sum" in plan, plan
    o, _ = _filtered_echo_graph(orc, noise, delays, gains, variant, out_channels)
    scale = max(1.0, float(np.abs(o).max()))
    assert rms_err(ring, o).max() <= 1e-6 * scale
    assert np.abs(ring - o).max() <= 4e-6 * scale
    monkeypatch.setenv("WAA_NO_ECHO_BQ", "1")
    plain, plan = _filtered_echo_graph(hip, noise, delays, gains, variant, out_channels)
    assert "with the Biquad between" not in plan
    assert np.abs(plain - ring).max() <= 4e-6 * scale


SHORT_DELAYS = (np.float64([266, 300.5, 511.25, 777, 1031, 2040]) / 48000.0).astype(np.float32)   # all below one 2048-frame tile
TINY_DELAYS = (np.float64([137, 150.5, 200.25, 240, 263, 1500]) / 48000.0).astype(np.float32)     # ... and below a 256-frame chunk + 8


@pytest.mark.measure
@pytest.mark.gpu
@pytest.mark.parametrize("channels,out_channels", [(1, 2), (2, 2)])
@pytest.mark.parametrize("variant", ["dry+wet", "wet-only", "wet-gain", "two-readers", "two-sources"])
def test_parity_short_echo_loop_from_the_lds_ring(hip, orc, channels, out_channels, variant, monkeypatch):
    """feedback delays of a few hundred frames (comb filters, plucked strings): shorter than a tile, so launches per block cannot
    render them and they used to go to the quantum-serial loop kernel; the ring kernel walks them in chunks of 256 frames with a
    ring as small as the delays need.  Bit-identical to the oracle and to the loop kernel (WAA_NO_SHORT_RING)"""
    n, frames = 6, 2048 * 5 + 77
    noise = white_noise(n, channels, frames, seed0=41)
    gains = np.float32([0.5, -0.7, 0.9, 0.3, 0.6, -0.95])
    ring, plan = _echo_graph(hip, noise, SHORT_DELAYS, gains, variant, out_channels)
    assert "LDS-ring kernel in ONE launch" in plan and "chunks of 256 frames" in plan and "shorter than a tile" in plan, plan
    assert "the line's last 4096 frames stay in LDS" in plan   # (2040 + 256 + 8 frames -> 4096)
    o, _ = _echo_graph(orc, noise, SHORT_DELAYS, gains, variant, out_channels)
    tol = 2e-6 if variant == "two-readers" else 0.0
    assert np.abs(ring - o).max() <= tol
    monkeypatch.setenv("WAA_NO_SHORT_RING", "1")
    serial, plan = _echo_graph(hip, noise, SHORT_DELAYS, gains, variant, out_channels)
    assert "LDS-ring kernel" not in plan
    assert np.abs(serial - ring).max() <= tol


@pytest.mark.gpu
@pytest.mark.parametrize("filtered", [False, True])
def test_parity_tiny_echo_loops_walk_in_half_wave_chunks(hip, orc, filtered):
    """delays of 137 .. 263 frames (a plucked string above ~180 Hz): chunks of 128 frames, the upper half of the wavefront idle"""
    n, frames = 6, 2048 * 3 + 77
    noise = white_noise(n, 2, frames, seed0=44)
    gains = np.float32([0.5, -0.7, 0.9, 0.3, 0.6, -0.95])
    graph = _filtered_echo_graph if filtered else _echo_graph
    g, plan = graph(hip, noise, TINY_DELAYS, gains, "dry+wet")
    assert "LDS-ring kernel in ONE launch" in plan and "chunks of 128 frames" in plan and "shorter than a tile" in plan, plan
    o, _ = graph(orc, noise, TINY_DELAYS, gains, "dry+wet")
    if filtered:
        assert rms_err(g, o).max() <= 1e-6 and np.abs(g - o).max() <= 4e-6
    else:
        assert np.array_equal(g, o)


@pytest.mark.gpu
@pytest.mark.parametrize("delay_frames", [144.0, 300.0])
def test_parity_plucked_string_with_a_long_memory_filter(hip, orc, delay_frames):
    """a 90 Hz lowpass (Q 3) in a short loop: the filter's state carries far across the ring kernel's chunks — with half-wave chunks
    (128 frames) the power that carries it across one sub-tile is A^32, not A^64 (fuzz seed 661385: 4e-4); mono line, stored for a
    reader outside the loop"""
    n, frames = 3, 2048 * 4 + 77
    noise = white_noise(n, 1, frames, seed0=45)
    outs = []
    for be in (hip, orc):
        c = waa.OfflineAudioContext(2, frames, 48000.0, n_instances=n, binding=be)
        src = c.create_buffer_source()
        src.set_buffer_batch(noise, 48000.0)
        delay = c.create_delay(0.1, delay_time=delay_frames / 48000.0)
        bq = c.create_biquad_filter(type_="lowpass", frequency=90.0, q=3.0)
        fb = c.create_gain(gain=0.8)
        src.connect(delay)
        delay.connect(bq).connect(fb).connect(delay)
        delay.connect(c.create_gain(gain=0.5)).connect(c.destination())   # (the line has a reader outside the loop)
        bq.connect(c.destination())
        src.start()
        if be is hip:
            plan = c.plan_describe()
            assert "with the Biquad between the delayed read and the sum" in plan and ("chunks of 128 frames" in plan) == (delay_frames < 264), plan
        outs.append(c.start_rendering_sync().data)
        c.close()
    g, o = outs
    scale = max(1.0, float(np.abs(o).max()))
    assert rms_err(g, o).max() <= 1e-6 * scale and np.abs(g - o).max() <= 4e-6 * scale, (rms_err(g, o), float(np.abs(g - o).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["dry+wet", "wet-gain", "line-reader", "peaking"])
def test_parity_short_filtered_echo_loop_from_the_lds_ring(hip, orc, variant):
    """the plucked string: Delay (a few hundred frames) -> Biquad -> Gain -> back, in the ring kernel's BQ form"""
    n, frames = 6, 2048 * 5 + 77
    noise = white_noise(n, 2, frames, seed0=42)
    gains = np.float32([0.5, -0.7, 0.9, 0.3, 0.6, -0.95])
    ring, plan = _filtered_echo_graph(hip, noise, SHORT_DELAYS, gains, variant)
    assert "with the Biquad between the delayed read and the sum" in plan and "chunks of 256 frames" in plan and "shorter than a tile" in plan, plan
    o, _ = _filtered_echo_graph(orc, noise, SHORT_DELAYS, gains, variant)
    scale = max(1.0, float(np.abs(o).max()))
    assert rms_err(ring, o).max() <= 1e-6 * scale
    assert np.abs(ring - o).max() <= 4e-6 * scale


@pytest.mark.gpu
def test_short_loops_the_ring_kernel_does_not_render_fall_back(hip, orc):
    """one instance's delay below 136 frames (128-frame chunk + 8), or a member that is not Delay / Gain / constant Biquad: the second
    planning pass hands the loop to the quantum-serial kernel"""
    n, frames = 6, 2048 * 3 + 77
    noise = white_noise(n, 2, frames, seed0=43)
    gains = np.float32([0.5, -0.7, 0.9, 0.3, 0.6, -0.95])
    delays = SHORT_DELAYS.copy()
    delays[2] = np.float32(130.0 / 48000.0)
    g, plan = _echo_graph(hip, noise, delays, gains, "dry+wet")
    assert "LDS-ring kernel" not in plan
    assert np.abs(g - _echo_graph(orc, noise, delays, gains, "dry+wet")[0]).max() == 0.0

    def shaped(binding):
        c = waa.OfflineAudioContext(2, frames, 48000.0, n_instances=n, binding=binding)
        src = c.create_buffer_source()
        src.set_buffer_batch(noise, 48000.0)
        delay = c.create_delay(0.4, delay_time=0.01)
        src.connect(delay)
        delay.connect(c.create_wave_shaper(curve=np.float32([-0.8, 0.0, 0.8]))).connect(c.create_gain(gain=0.6)).connect(delay)
        delay.connect(c.destination())
        src.start()
        plan = c.plan_describe() if binding.prefix == "waa_" else ""
        out = c.start_rendering_sync().data
        c.close()
        return out, plan
    g, plan = shaped(hip)
    assert "LDS-ring kernel" not in plan
    assert np.abs(g - shaped(orc)[0]).max() <= 1e-6
