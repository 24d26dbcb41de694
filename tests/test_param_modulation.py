"""Audio-rate modulation of AudioParams from the graph (`node.connect(&param)`, src/param.rs:686-795; SURVEY.md
§8f rank 2).  The reference's own coverage: tests/offline.rs (param modulated by a constant source), param.rs unit
tests of mix_to_output; re-typed here, plus GPU-vs-oracle parity on seeded inputs."""
import numpy as np
import pytest

import web_audio_api_rs_amd as waa
from graphs import rms_err, white_noise

RQ = 128


def ctx(be, channels, length, sr=48000.0, **kw):
    return waa.OfflineAudioContext(channels, length, sr, binding=be, **kw)


def test_constant_source_into_gain_param(be):
    """param.rs:737-770: the input is ADDED to the intrinsic value (gain 1 + offset 0.5), result clamped"""
    c = ctx(be, 1, RQ * 3)
    src = c.create_constant_source(offset=2.0)
    g = c.create_gain(gain=1.0)
    mod = c.create_constant_source(offset=0.5)
    src.connect(g).connect(c.destination())
    mod.connect(g.gain)
    src.start()
    mod.start_at(RQ / 48000.0)  # modulator silent during the first quantum
    out = c.start_rendering_sync().data[0, 0]
    assert np.allclose(out[:RQ], 2.0, atol=0) and np.allclose(out[RQ:], 3.0, atol=0)


def test_param_input_is_discrete_channel_zero_and_clamped(be):
    """param.rs:309-311: channel count 1 / explicit / discrete -> only channel 0 of a stereo modulator counts;
    param.rs:755-761: NaN -> default, then max/min clamp (StereoPanner pan in [-1, 1])"""
    sr = 48000.0
    c = ctx(be, 2, RQ)
    src = c.create_constant_source(offset=1.0)
    pan = c.create_stereo_panner(pan=0.0)
    modbuf = np.zeros((2, RQ), np.float32)
    modbuf[0, :64] = 5.0     # clamps to pan = +1: everything right
    modbuf[0, 64:] = -5.0    # clamps to pan = -1: everything left
    modbuf[1, :] = 123.0     # ignored (discrete down-mix keeps channel 0)
    mod = c.create_buffer_source()
    mod.set_buffer(waa.AudioBuffer(modbuf, sr))
    src.connect(pan).connect(c.destination())
    mod.connect(pan.pan)
    src.start()
    mod.start()
    out = c.start_rendering_sync().data[0]
    assert np.abs(out[0, :64]).max() <= 1e-6 and np.allclose(out[1, :64], 1.0, atol=1e-6)
    assert np.allclose(out[0, 64:], 1.0, atol=1e-6) and np.abs(out[1, 64:]).max() <= 1e-6


def test_two_modulators_sum(be):
    c = ctx(be, 1, RQ)
    src = c.create_constant_source(offset=1.0)
    g = c.create_gain(gain=0.0)
    a = c.create_constant_source(offset=0.25)
    b2 = c.create_constant_source(offset=0.5)
    src.connect(g).connect(c.destination())
    a.connect(g.gain)
    b2.connect(g.gain)
    for n in (src, a, b2):
        n.start()
    out = c.start_rendering_sync().data[0, 0]
    assert np.allclose(out, 0.75, atol=0)


def test_a_constant_on_a_panner_position_equals_the_moved_position(be):
    """a graph input on a PannerNode param is ADDED to its value (param.rs:739-795) — rendered since round 4 (the node's own six
    params with a single-valued listener; the oracle everywhere): a ConstantSource of 2 on positionX of a panner at x = 1 is the
    panner at x = 3"""
    outs = []
    for moved in (False, True):
        c = ctx(be, 2, RQ * 6)
        src = c.create_buffer_source()
        src.set_buffer(waa.AudioBuffer(np.linspace(-1, 1, RQ * 6, dtype=np.float32)[None, :], 48000.0))
        pan = c.create_panner(position=(3.0 if moved else 1.0, 0.0, -1.0))
        if not moved:
            k = c.create_constant_source(offset=2.0)
            k.connect(pan.position_x)
            k.start()
        src.connect(pan).connect(c.destination())
        src.start()
        outs.append(c.start_rendering_sync().data)
    assert np.array_equal(outs[0], outs[1]) and np.abs(outs[0]).max() > 0.1


def test_plan_param_chain_precedes_consumer(hip):
    c = waa.OfflineAudioContext(2, RQ * 32, 48000.0, n_instances=2, binding=hip, device=waa.PLAN_ONLY)
    src = c.create_buffer_source()
    src.set_buffer_batch(white_noise(2, 2, RQ * 32), 48000.0)
    lfo = c.create_buffer_source()
    lfo.set_buffer_batch(white_noise(2, 1, RQ * 32, seed0=5), 48000.0)
    g = c.create_gain(gain=0.5)
    src.connect(g).connect(c.destination())
    lfo.connect(g.gain)
    src.start()
    lfo.start()
    plan = c.plan_describe().splitlines()
    i_param = next(i for i, l in enumerate(plan) if "PARAM_ADD" in l)
    i_gain = next(i for i, l in enumerate(plan) if "GAIN" in l)
    assert i_param < i_gain
    c.close()


# --------------------------------------------------------------------------- GPU parity
def _am_tremolo(binding, noise, lfo):
    n = noise.shape[0]
    c = waa.OfflineAudioContext(2, noise.shape[2], 48000.0, n_instances=n, binding=binding)
    src = c.create_buffer_source()
    src.set_buffer_batch(noise, 48000.0)
    mod = c.create_buffer_source()
    mod.set_buffer_batch(lfo, 48000.0)
    depth = c.create_gain(gain=0.4)
    g = c.create_gain(gain=0.6)
    src.connect(g).connect(c.destination())
    mod.connect(depth).connect(g.gain)
    src.start()
    mod.start()
    out = c.start_rendering_sync().data
    c.close()
    return out


@pytest.mark.gpu
def test_parity_am_tremolo(hip, orc):
    n, frames = 4, 2048 * 2 + 77
    noise = white_noise(n, 2, frames, seed0=1)
    t = np.arange(frames) / 48000.0
    lfo = np.stack([np.sin(2 * np.pi * (3.0 + i) * t) for i in range(n)]).astype(np.float32)[:, None, :]
    g, o = _am_tremolo(hip, noise, lfo), _am_tremolo(orc, noise, lfo)
    assert np.abs(g - o).max() <= 1e-7


@pytest.mark.gpu
def test_parity_filter_sweep_and_flanger(hip, orc):
    """an LFO buffer drives Biquad.frequency (a-rate coefficients) and Delay.delayTime (a-rate gather)"""
    n, frames = 3, 2048 * 2
    noise = white_noise(n, 2, frames, seed0=2)
    t = np.arange(frames) / 48000.0
    lfo = np.stack([np.sin(2 * np.pi * (1.0 + i) * t) for i in range(n)]).astype(np.float32)[:, None, :]
    outs = []
    for be_ in (hip, orc):
        c = waa.OfflineAudioContext(2, frames, 48000.0, n_instances=n, binding=be_)
        src = c.create_buffer_source()
        src.set_buffer_batch(noise, 48000.0)
        mod = c.create_buffer_source()
        mod.set_buffer_batch(lfo, 48000.0)
        bq = c.create_biquad_filter(type_="lowpass", frequency=1200.0, q=2.0)
        fdepth = c.create_gain(gain=800.0)
        dl = c.create_delay(0.02, delay_time=0.005)
        ddepth = c.create_gain(gain=0.003)
        src.connect(bq).connect(dl).connect(c.destination())
        src.connect(c.destination())
        mod.connect(fdepth).connect(bq.frequency)
        mod.connect(ddepth).connect(dl.delay_time)
        src.start()
        mod.start()
        outs.append(c.start_rendering_sync().data)
        c.close()
    assert rms_err(*outs).max() <= 1e-6
    assert np.abs(outs[0] - outs[1]).max() <= 2e-6


# ---- playbackRate / detune of an AudioBufferSourceNode modulated from the graph (k-rate params: param.rs:739-760,
# audio_buffer_source.rs:176-197).  The host replays the playhead, so the library renders the modulating subgraph at plan
# time and reads one value per render quantum back (waa_abi.cpp::resolve_source_rate_modulation).
def _ramp_buffer(frames, n_ch=1):
    t = np.arange(frames, dtype=np.float32)
    return np.stack([np.sin(t * (0.01 + 0.003 * c)) for c in range(n_ch)]).astype(np.float32)


def _render_rate(be, frames, modulate, rate=1.0, extra=0.5, start_mod=0.0, device=-1):
    c = ctx(be, 1, frames) if device == -1 else waa.OfflineAudioContext(1, frames, 48000.0, binding=be, device=device)
    src = c.create_buffer_source()
    src.set_buffer(waa.AudioBuffer(_ramp_buffer(frames * 3), 48000.0))
    src.playback_rate.set_value(rate)
    if modulate:
        mod = c.create_constant_source(offset=extra)
        mod.connect(src.playback_rate)
        mod.start_at(start_mod)
    src.connect(c.destination())
    src.start()
    return c


def test_constant_modulator_on_playback_rate_equals_the_sum(be):
    """rate 1.0 + a ConstantSource of 0.5 on the param = playbackRate 1.5 (the input is ADDED, once per quantum)"""
    frames = RQ * 24
    a = _render_rate(be, frames, True, 1.0, 0.5).start_rendering_sync().data
    c = _render_rate(be, frames, False, 1.5)
    b = c.start_rendering_sync().data
    assert np.abs(a).max() > 0.5 and np.array_equal(a, b)


def test_modulator_that_starts_later_changes_the_rate_at_a_quantum_boundary(be):
    """k-rate: the FIRST sample of the quantum counts — a modulator that starts in the middle of quantum 3 (its first
    frame still 0 there) moves the rate from quantum 4 on"""
    frames = RQ * 16
    start = (3 * RQ + 40) / 48000.0
    out = _render_rate(be, frames, True, 1.0, 1.0, start_mod=start).start_rendering_sync().data[0, 0]
    plain = _render_rate(be, frames, False, 1.0).start_rendering_sync().data[0, 0]
    assert np.array_equal(out[:4 * RQ], plain[:4 * RQ])          # quanta 0..3 at rate 1
    buf = _ramp_buffer(frames * 3)[0]
    # from quantum 4 on the playhead advances 2 buffer frames per frame, starting where rate 1 left it
    expect = buf[4 * RQ + 2 * np.arange(RQ)]
    assert np.abs(out[4 * RQ:5 * RQ] - expect).max() <= 1e-6


def test_plan_only_batch_names_the_reason(hip):
    c = _render_rate(hip, RQ * 8, True, device=waa.PLAN_ONLY)
    with pytest.raises(waa.WaaError) as ei:
        c.plan_describe()
    assert ei.value.status == 4 and "rendered at plan time" in str(ei.value)
    c.close()


def _vibrato(binding, noise, lfo_hz, depth, detune_cents, n_ch):
    """the vibrato patch: Oscillator(LFO) -> Gain(depth) -> source.playbackRate, a second LFO on detune"""
    n, _, frames = noise.shape
    c = waa.OfflineAudioContext(n_ch, frames // 2, 48000.0, n_instances=n, binding=binding)
    src = c.create_buffer_source()
    src.set_buffer_batch(noise, 48000.0)
    src.set_loop(True)
    lfo = c.create_oscillator(type_="sine", frequency=lfo_hz)
    for i in range(n):
        lfo.frequency.set_value(lfo_hz * (1.0 + 0.3 * i), instance=i)
    d = c.create_gain(gain=depth)
    lfo.connect(d).connect(src.playback_rate)
    if detune_cents:
        lfo2 = c.create_oscillator(type_="triangle", frequency=lfo_hz * 0.37)
        d2 = c.create_gain(gain=detune_cents)
        lfo2.connect(d2).connect(src.detune)
        lfo2.start()
    src.connect(c.create_gain(gain=0.5)).connect(c.destination())
    lfo.start()
    src.start()
    return c


@pytest.mark.gpu
@pytest.mark.parametrize("detune_cents", [0.0, 300.0])
def test_parity_vibrato_on_playback_rate(hip, orc, detune_cents):
    """per-instance LFO rates: every instance gets its own playhead schedule (the slow, interpolating track)"""
    noise = white_noise(3, 2, RQ * 160, seed0=77)
    outs = []
    for be in (hip, orc):
        c = _vibrato(be, noise, 6.0, 0.2, detune_cents, 2)
        if be is hip:
            assert "modulated from the graph" in c.plan_describe()
        outs.append(c.start_rendering_sync().data)
        c.close()
    g, o = outs
    assert np.abs(o).max() > 0.1
    assert rms_err(g, o).max() <= 1e-6 and np.abs(g - o).max() <= 2e-5


@pytest.mark.gpu
def test_modulated_rate_source_next_to_a_plain_one(hip, orc):
    """the rest of the graph plans as usual once the param edges are resolved: a filter behind the modulated source, a second
    plain source, both into the destination"""
    noise = white_noise(2, 1, RQ * 120, seed0=5)
    outs = []
    for be in (hip, orc):
        c = _vibrato(be, noise, 4.0, 0.5, 0.0, 1)
        plain = c.create_buffer_source()
        plain.set_buffer_batch(noise[:, :, ::-1].copy(), 48000.0)
        plain.connect(c.create_biquad_filter(type_="bandpass", frequency=1500.0)).connect(c.destination())
        plain.start()
        outs.append(c.start_rendering_sync().data)
        c.close()
    assert rms_err(outs[0], outs[1]).max() <= 1e-6


# ---- PannerNode position / orientation driven by the graph (round 4) -------------------------------------------------
def _orbiting_panner(binding, noise, lfo_hz, model, device=-1, a_rate_listener=False):
    """a source that circles the listener: two LFOs (sine / cosine-ish: a triangle a quarter period late) -> Gain(radius) ->
    panner.positionX / positionZ, a third one wobbles orientationX; per-instance LFO rates"""
    n, n_ch, frames = noise.shape
    c = waa.OfflineAudioContext(2, frames, 48000.0, n_instances=n, binding=binding, device=device)
    src = c.create_buffer_source()
    src.set_buffer_batch(noise, 48000.0)
    pan = c.create_panner(panning_model=model, distance_model="inverse", position=(0.0, 0.5, -1.0), ref_distance=1.0,
                          cone_inner_angle=60.0, cone_outer_angle=200.0, cone_outer_gain=0.3)
    lx = c.create_oscillator(type_="sine", frequency=lfo_hz)
    lz = c.create_oscillator(type_="triangle", frequency=lfo_hz)
    lo = c.create_oscillator(type_="sine", frequency=lfo_hz * 2.3)
    for i in range(n):
        lx.frequency.set_value(lfo_hz * (1.0 + 0.4 * i), instance=i)
        lz.frequency.set_value(lfo_hz * (1.0 + 0.4 * i), instance=i)
    lx.connect(c.create_gain(gain=3.0)).connect(pan.position_x)
    lz.connect(c.create_gain(gain=2.0)).connect(pan.position_z)
    lo.connect(c.create_gain(gain=0.8)).connect(pan.orientation_x)
    pan.position_y.set_value_at_time(0.5, 0.0).linear_ramp_to_value_at_time(2.0, frames / 48000.0)  # automation next to it
    if a_rate_listener:
        c.listener().position_x.set_value_at_time(0.0, 0.0).linear_ramp_to_value_at_time(1.0, frames / 48000.0)
    src.connect(pan).connect(c.destination())
    lx.start()
    lz.start_at(0.013)   # (starts mid-quantum: its first value is seen at the NEXT quantum boundary)
    lo.start()
    src.start()
    return c


@pytest.mark.gpu
@pytest.mark.parametrize("model,n_ch", [("equalpower", 1), ("equalpower", 2), ("HRTF", 1)])
def test_parity_panner_position_modulated_from_the_graph(hip, orc, model, n_ch):
    """panner.rs:833-846 (HRTF: :781-829): with a single-valued AudioListener the node uses the FIRST value of every param per
    render quantum — graph input included (param.rs:739-795).  The modulating subgraph is rendered at plan time, one value per
    quantum read back (the mechanism of the graph-modulated playbackRate), then the ordinary plan."""
    noise = white_noise(3, n_ch, RQ * 150, seed0=91)
    outs = []
    for be in (hip, orc):
        c = _orbiting_panner(be, noise, 3.0, model)
        if be is hip:
            assert "modulated from the graph" in c.plan_describe() and "panner" in c.plan_describe()
        outs.append(c.start_rendering_sync().data)
        c.close()
    g, o = outs
    assert np.abs(o).max() > 0.05
    assert rms_err(g, o).max() <= 1e-6 and np.abs(g - o).max() <= 2e-5


def test_modulated_panner_geometry_needs_the_device_and_a_single_valued_listener(hip):
    noise = white_noise(2, 1, RQ * 8)
    c = _orbiting_panner(hip, noise, 3.0, "equalpower", device=waa.PLAN_ONLY)
    with pytest.raises(waa.WaaError) as ei:
        c.plan_describe()
    assert ei.value.status == 4 and "resolved at plan time" in str(ei.value)
    c.close()


@pytest.mark.gpu
def test_modulated_panner_geometry_next_to_an_audio_rate_listener_is_refused(hip):
    c = _orbiting_panner(hip, white_noise(2, 1, RQ * 8), 3.0, "equalpower", a_rate_listener=True)
    with pytest.raises(waa.WaaError) as ei:
        c.start_rendering_sync()
    assert ei.value.status == 4 and "single-valued AudioListener" in str(ei.value)
    c.close()


# ---- an OSCILLATOR as the LFO (round 4: the param's summing chain is folded into the oscillator's store) -----------------------
def _osc_lfo(binding, noise, variant, plan_only=False):
    """tremolo       source -> Gain <- (sine LFO -> depth Gain) on gain.gain
    sweep         the LFO (triangle, per-instance rate, starting late) -> depth -> Biquad.frequency
    no-depth      the LFO straight on gain.gain (no depth gain), intrinsic value automated per... constant per instance
    two-params    ONE LFO drives two params (its signal has two readers: not folded)
    heard         the LFO also reaches the destination (not folded)"""
    n, frames = noise.shape[0], noise.shape[2]
    kw = {"device": waa.PLAN_ONLY} if plan_only else {}
    c = waa.OfflineAudioContext(2, frames, 48000.0, n_instances=n, binding=binding, **kw)
    src = c.create_buffer_source()
    src.set_buffer_batch(noise, 48000.0)
    lfo = c.create_oscillator(type_="triangle" if variant == "sweep" else "sine", frequency=4.0)
    for i in range(n):
        lfo.frequency.set_value(3.0 + 1.5 * i, instance=i)
    g = c.create_gain(gain=0.6)
    for i in range(n):
        g.gain.set_value(0.5 + 0.1 * i, instance=i)
    bq = c.create_biquad_filter(type_="lowpass", frequency=1500.0, q=3.0)
    if variant == "sweep":
        lfo.connect(c.create_gain(gain=900.0)).connect(bq.frequency)
        src.connect(bq).connect(c.destination())
        lfo.start_at(300.5 / 48000.0)
    else:
        head = lfo if variant == "no-depth" else lfo.connect(c.create_gain(gain=0.4))
        head.connect(g.gain)
        if variant == "two-params":
            head.connect(bq.detune)
        if variant == "heard":
            lfo.connect(c.create_gain(gain=0.05)).connect(c.destination())
        src.connect(g).connect(bq).connect(c.destination())
        lfo.start()
    src.start()
    plan = c.plan_describe() if binding.prefix == "waa_" else ""
    out = None if plan_only else c.start_rendering_sync().data
    c.close()
    return out, plan


LFO_VARIANTS = ["tremolo", "sweep", "no-depth", "two-params", "heard"]


@pytest.mark.measure
@pytest.mark.parametrize("variant", LFO_VARIANTS)
def test_plan_lfo_fold(variant):
    _, plan = _osc_lfo(waa.measure_binding(), white_noise(3, 2, 128 * 40), variant, plan_only=True)
    assert ("is folded into the store of the oscillator that drives it" in plan) == (variant not in ("two-params", "heard")), plan


@pytest.mark.measure
@pytest.mark.gpu
@pytest.mark.parametrize("variant", LFO_VARIANTS)
def test_parity_lfo_fold(hip, orc, variant, monkeypatch):
    """the oscillator's store applying depth gain, intrinsic value and clamp is the chain launch's arithmetic: bit-identical to the
    unfolded plan (WAA_NO_LFO_FOLD), and the render matches the oracle"""
    noise = white_noise(3, 2, 2048 * 3 + 77, seed0=9)
    g, plan = _osc_lfo(hip, noise, variant)
    assert ("is folded into the store of the oscillator that drives it" in plan) == (variant not in ("two-params", "heard")), plan
    o, _ = _osc_lfo(orc, noise, variant)
    assert rms_err(g, o).max() <= 1e-6 and np.abs(g - o).max() <= 4e-6
    monkeypatch.setenv("WAA_NO_LFO_FOLD", "1")
    u, plan = _osc_lfo(hip, noise, variant)
    assert "is folded into the store" not in plan
    assert np.array_equal(g, u)


def _feedback_gain_loop(be, noise, delay_s, variant, gain0=0.2, depth=0.3):
    """src -> Delay -> Gain(g) -> back into the Delay, and g.gain is driven by the loop's own signal:
    direct     Delay -> g.gain                       (the reader's output of this quantum)
    depth      Delay -> Gain(depth) -> g.gain        (through a member of the loop)
    two        Delay -> g.gain  and  Delay -> Gain(depth) -> g.gain   (two inputs, summed in edge order)"""
    n = noise.shape[0]
    c = waa.OfflineAudioContext(2, noise.shape[2], 48000.0, n_instances=n, binding=be)
    src = c.create_buffer_source()
    src.set_buffer_batch(noise, 48000.0)
    d = c.create_delay(1.0, delay_time=delay_s)
    g = c.create_gain(gain=gain0)
    for i in range(n):
        g.gain.set_value(gain0 + 0.05 * i, instance=i)
    src.connect(d)
    d.connect(g).connect(d)
    if variant in ("direct", "two"):
        d.connect(g.gain)
    if variant in ("depth", "two"):
        dp = c.create_gain(gain=depth)
        d.connect(dp)
        dp.connect(g.gain)
    g.connect(c.destination())
    src.connect(c.destination())
    src.start()
    return c


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["direct", "depth", "two"])
@pytest.mark.parametrize("delay_s", [0.0, 0.004, 0.031])
def test_parity_gain_modulated_from_inside_its_own_loop(hip, orc, delay_s, variant):
    """round 6: an AudioParam driven by a member of the node's own feedback loop (legal: graph.rs:340-361 orders param edges like any
    other, param.rs:762-795 sums them per quantum) was status 4 for quantum-serial loops.  A GainNode's gain is now rendered by the
    dynamic-count kernel itself: the param's inputs are the loop members' outputs of the same quantum (channel 0: count 1, explicit,
    discrete), plus the intrinsic value, clamped.  (Loops with block-long delays rendered it before: node-major param chains.)"""
    noise = white_noise(3, 2, 60 * RQ) * 0.1  # (small: the loop gain follows the signal, |g| stays well below 1)
    outs = []
    for be in (hip, orc):
        c = _feedback_gain_loop(be, noise, delay_s, variant)
        if be is hip:
            assert "GAIN" in c.plan_describe()
        outs.append(c.start_rendering_sync().data)
        c.close()
    g, o = outs
    assert np.isfinite(o).all() and np.abs(o).max() < 50.0
    x = white_noise(3, 2, 60 * RQ) * 0.1
    assert np.abs(o - x).max() > 0.01  # (the loop contributes)
    assert rms_err(g, o).max() <= 1e-6 * max(1.0, float(np.abs(o).max()))


def test_a_modulated_gain_that_is_zero_while_its_modulator_is_silent_is_refused(hip):
    """suspend fuzz seed 90491 (round 6).  While the audio-rate input of a GainNode's gain is silent the param is ONE value per quantum
    (param.rs:737-760) and the node's fast paths apply: a gain of 0 emits the silent block — one channel —, the modulated kernels
    multiply by a per-frame table of zeros and keep the static layout.  Not rendered differently: status 4 with the quantum"""
    def build(zero_at, stop):
        c = waa.OfflineAudioContext(2, 40 * RQ, 48000.0, n_instances=2, binding=hip, device=waa.PLAN_ONLY)
        src = c.create_buffer_source()
        src.set_buffer_batch(white_noise(2, 2, 40 * RQ), 48000.0)
        src.start()
        g = c.create_gain(gain=0.5)
        vals = np.full(40, 0.5, np.float32)
        if zero_at is not None:
            vals[zero_at] = 0.0
        g.gain.set_block(0, vals)
        lfo = c.create_oscillator(type_="sine", frequency=7.0)
        depth = c.create_gain(gain=0.2)
        lfo.connect(depth)
        depth.connect(g.gain)
        lfo.start()
        if stop is not None:
            lfo.stop_at(stop * RQ / 48000.0)
        src.connect(g).connect(c.create_stereo_panner(pan=0.3)).connect(c.destination())
        return c
    for zero_at, stop in ((None, 20), (30, None), (10, 20)):  # no zero / the modulator never stops / the zero falls while it still runs
        c = build(zero_at, stop)
        assert "PARAM_ADD" in c.plan_describe()
        c.close()
    c = build(30, 20)
    with pytest.raises(waa.WaaError) as ei:
        c.plan_describe()
    assert ei.value.status == 4 and "gain is 0 in quantum 30" in str(ei.value)
    c.close()
