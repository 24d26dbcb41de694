"""OfflineAudioContext::suspend_sync and the ranged render of the C ABI (round 6; waa_render_range / waa_connect / waa_disconnect).

The reference pauses at render-quantum boundaries, runs a callback that may edit the graph, handles the control messages the
callback submitted and renders on (src/context/offline.rs:359-397, src/render/thread.rs:277-294).  The tests of
src/context/offline.rs:469-575 are re-typed here for both back-ends; then graphs are mutated at suspend points — a gain value, a
connection, automation, start / stop — and the device render is held against the oracle, which applies every control message in
front of its quantum in a real quantum loop (the product library compiles the history into its plan instead: gated connections,
clamped start times, late automation events — two independent statements of the same semantics)."""
import numpy as np
import pytest

import web_audio_api_rs_amd as waa
from graphs import white_noise

RQ = 128
SR = 48000.0


def rms_err(a, b):
    return np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2, axis=-1))


# ---- the reference's own tests (offline.rs:469-575), on the oracle (CPU) and on the device ---------------------------------------

def _test_suspend_sync(binding):
    length = RQ * 4
    ctx = waa.OfflineAudioContext(1, length, SR, binding=binding)
    box = {}

    def at_one(c):
        assert c.state() == "suspended"
        src = c.create_constant_source()
        src.connect(c.destination())
        src.start()
        box["src"] = src

    def at_three(c):
        assert c.state() == "suspended"
        box["src"].disconnect()

    ctx.suspend_sync(RQ / SR, at_one)
    ctx.suspend_sync(3 * RQ / SR, at_three)
    out = ctx.start_rendering_sync().get_channel_data(0)
    assert np.array_equal(out[:RQ], np.zeros(RQ, np.float32))
    assert np.array_equal(out[RQ:3 * RQ], np.ones(2 * RQ, np.float32))
    assert np.array_equal(out[3 * RQ:], np.zeros(RQ, np.float32))
    assert ctx.state() == "closed"
    ctx.close()


def test_suspend_sync_oracle(orc):
    _test_suspend_sync(orc)


@pytest.mark.gpu
def test_suspend_sync_device(hip):
    _test_suspend_sync(hip)


def test_suspend_negative_panics(orc):
    ctx = waa.OfflineAudioContext(2, 128, 44100.0, binding=orc)
    with pytest.raises(waa.WaaError, match="suspendTime cannot be negative"):
        ctx.suspend_sync(-1.0, lambda c: None)


def test_suspend_after_duration_panics(orc):
    ctx = waa.OfflineAudioContext(2, 128, 44100.0, binding=orc)
    with pytest.raises(waa.WaaError, match="greater than or equal to the total render duration"):
        ctx.suspend_sync(1.0, lambda c: None)


def test_suspend_after_render_panics(orc):
    ctx = waa.OfflineAudioContext(2, 128, 44100.0, binding=orc)
    ctx.start_rendering_sync()
    with pytest.raises(waa.WaaError, match="cannot suspend when rendering has already started"):
        ctx.suspend_sync(0.0, lambda c: None)
    ctx.close()


def test_suspend_identical_frame_panics(orc):
    ctx = waa.OfflineAudioContext(2, 128, 44100.0, binding=orc)
    ctx.suspend_sync(0.0, lambda c: None)
    with pytest.raises(waa.WaaError, match="cannot suspend multiple times at the same render quantum"):
        ctx.suspend_sync(0.0, lambda c: None)


def test_suspend_time_is_quantised_up(orc):
    """calculate_suspend_frame (offline.rs:241-251): ceil(time * rate / 128) — a suspend inside quantum 1 runs in front of quantum 2"""
    ctx = waa.OfflineAudioContext(1, RQ * 4, SR, binding=orc)

    def cb(c):
        src = c.create_constant_source()
        src.connect(c.destination())
        src.start()

    ctx.suspend_sync((RQ + 1) / SR, cb)
    out = ctx.start_rendering_sync().get_channel_data(0)
    assert not out[:2 * RQ].any() and np.all(out[2 * RQ:] == 1.0)
    ctx.close()


# ---- the ABI itself ----------------------------------------------------------------------------------------------------------

def _raw_batch(binding, n_quanta=6):
    import ctypes as C
    ctx = waa.OfflineAudioContext(1, RQ * n_quanta, SR, binding=binding)
    src = ctx.create_constant_source()
    gain = ctx.create_gain(gain=0.5)
    src.connect(gain)
    src.start()
    ctx.prepare()
    return ctx, src, gain, C


@pytest.mark.parametrize("which", ["orc", "hip_plan_only"])
def test_render_range_rules(orc, hip_product, which):
    """ranges are consecutive from quantum 0, stay inside the render, and nothing can be read before the last one"""
    binding = orc if which == "orc" else hip_product
    kw = {} if which == "orc" else {"device": waa.PLAN_ONLY}
    ctx = waa.OfflineAudioContext(1, RQ * 6, SR, binding=binding, **kw)
    src = ctx.create_constant_source()
    gain = ctx.create_gain(gain=0.5)
    src.connect(gain)
    src.start()
    ctx.prepare()
    b, h = ctx._b, ctx._handle
    with pytest.raises(waa.WaaError, match="ranges are consecutive"):
        b.check(b.render_range(h, 1, 2))
    with pytest.raises(waa.WaaError, match="RangeError"):
        b.check(b.render_range(h, 0, 7))
    with pytest.raises(waa.WaaError, match="RangeError"):
        b.check(b.render_range(h, 0, 0))
    b.check(b.render_range(h, 0, 2))
    with pytest.raises(waa.WaaError, match="ranges are consecutive"):
        b.check(b.render_range(h, 0, 2))
    # the graph is edited at the suspend point: gain -> destination from quantum 2 on
    b.check(b.connect(h, gain.id, 0, 0, 0))
    b.check(b.connect(h, gain.id, 0, 0, 0))  # twice: a no-op
    with pytest.raises(waa.WaaError, match="InvalidAccessError"):
        b.check(b.disconnect(h, src.id, 0, 0, 0))  # never connected
    with pytest.raises(waa.WaaError, match="IndexSizeError"):
        b.check(b.connect(h, 99, 0, 0, 0))
    out = np.zeros(RQ * 6, np.float32)
    with pytest.raises(waa.WaaError, match="InvalidStateError"):
        b.check(b.download(h, 0, 0, waa.api._fp(out), out.size))
    with pytest.raises(waa.WaaError, match="ranged render is in progress"):
        b.check(b.render(h))
    if which == "orc":
        b.check(b.render_range(h, 2, 4))
        b.check(b.download(h, 0, 0, waa.api._fp(out), out.size))
        assert not out[:2 * RQ].any() and np.all(out[2 * RQ:] == 0.5)
    else:
        text = ctx.plan_describe()
        assert "1 connection(s) made or cut at suspend points" in text
    ctx.close()


def test_connection_made_and_cut_at_the_same_point_never_exists(orc):
    ctx = waa.OfflineAudioContext(1, RQ * 4, SR, binding=orc)
    src = ctx.create_constant_source()
    src.start()

    def cb(c):
        src.connect(c.destination())
        src.disconnect(c.destination())

    ctx.suspend_sync(RQ / SR, cb)
    assert not ctx.start_rendering_sync().data.any()
    ctx.close()


# ---- graphs mutated at suspend points: the device against the oracle --------------------------------------------------------------

def _mutated_graph(binding, noise, n_quanta):
    """noise -> Biquad -> Gain -> destination and noise -> Delay (not connected yet).  Quantum 3: the gain jumps to 0.25 and the
    delay branch joins the destination; quantum 7: a linear ramp on the gain (scheduled from the callback: its start value is the
    value the timeline holds THEN), the biquad's Q changes; quantum 12: the filter branch is cut; quantum 15: delayTime changes."""
    n_inst = noise.shape[0]
    ctx = waa.OfflineAudioContext(2, RQ * n_quanta - 17, SR, n_instances=n_inst, binding=binding)
    src = ctx.create_buffer_source()
    src.set_buffer_batch(noise, SR)
    flt = ctx.create_biquad_filter(type_="lowpass", frequency=900.0, q=2.0)
    gain = ctx.create_gain(gain=0.8)
    dly = ctx.create_delay(0.5)
    dly.delay_time.set_value(0.004)
    src.connect(flt).connect(gain).connect(ctx.destination())
    src.connect(dly)
    src.start()

    def q3(c):
        gain.gain.set_value(0.25)
        dly.connect(c.destination())

    def q7(c):
        gain.gain.linear_ramp_to_value_at_time(1.0, 11 * RQ / SR)
        flt.q.set_value(0.7)

    def q12(c):
        flt.disconnect(gain)

    def q15(c):
        dly.delay_time.set_value(0.0095)

    for q, cb in ((3, q3), (7, q7), (12, q12), (15, q15)):
        if q < n_quanta:
            ctx.suspend_sync(q * RQ / SR, cb)
    return ctx


def test_mutated_graph_oracle_runs_and_changes_where_it_should(orc):
    noise = white_noise(2, 2, RQ * 20)
    ctx = _mutated_graph(orc, noise, 20)
    out = ctx.start_rendering_sync().data
    ctx.close()
    plain = waa.OfflineAudioContext(2, RQ * 20 - 17, SR, n_instances=2, binding=orc)
    src = plain.create_buffer_source()
    src.set_buffer_batch(noise, SR)
    flt = plain.create_biquad_filter(type_="lowpass", frequency=900.0, q=2.0)
    gain = plain.create_gain(gain=0.8)
    src.connect(flt).connect(gain).connect(plain.destination())
    src.start()
    ref = plain.start_rendering_sync().data
    plain.close()
    assert np.array_equal(out[:, :, :3 * RQ], ref[:, :, :3 * RQ])      # nothing happens before the first suspend point
    assert not np.array_equal(out[:, :, 3 * RQ:4 * RQ], ref[:, :, 3 * RQ:4 * RQ])
    assert np.all(np.isfinite(out))


@pytest.mark.gpu
def test_gain_and_connection_mutated_at_suspend_points_match_the_oracle(hip, orc):
    noise = white_noise(5, 2, RQ * 40)
    outs = []
    for binding in (hip, orc):
        ctx = _mutated_graph(binding, noise, 40)
        outs.append(ctx.start_rendering_sync().data)
        ctx.close()
    assert np.all(np.isfinite(outs[0])) and np.abs(outs[0]).max() > 1e-2
    assert rms_err(outs[0], outs[1]).max() <= 1e-6


def _sources_at_suspend(binding, noise):
    """sources created, started and stopped inside callbacks: a start time in the past (starts at the block), one in the future,
    a buffer source with an offset, a stop in the past (stops at the block), an oscillator"""
    n_inst = noise.shape[0]
    ctx = waa.OfflineAudioContext(2, RQ * 24, SR, n_instances=n_inst, binding=binding)
    box = {}

    def q2(c):
        s = c.create_buffer_source()
        s.set_buffer_batch(noise, SR)
        s.connect(c.destination())
        s.start_at_with_offset(0.0, 50 / SR)          # when = 0 has passed: plays from the block's first frame, buffer offset 50 frames
        box["buf"] = s
        k = c.create_constant_source()
        k.offset.set_value(0.125)
        k.connect(c.destination())
        k.start_at((5 * RQ + 37) / SR)                # in the future, mid-quantum
        box["const"] = k

    def q9(c):
        box["buf"].stop_at(1.0 / SR)                  # long past: stops at this block
        o = c.create_oscillator(type_="sawtooth", frequency=700.0)
        g = c.create_gain(gain=0.2)
        o.connect(g).connect(c.destination())
        o.start()
        box["osc"] = o

    def q14(c):
        box["const"].stop_at((16 * RQ + 5) / SR)
        box["osc"].frequency.set_value(1234.0)

    ctx.suspend_sync(2 * RQ / SR, q2)
    ctx.suspend_sync(9 * RQ / SR, q9)
    ctx.suspend_sync(14 * RQ / SR, q14)
    return ctx, box


def test_sources_started_inside_callbacks_oracle(orc):
    noise = white_noise(1, 2, RQ * 30)
    ctx, box = _sources_at_suspend(orc, noise)
    out = ctx.start_rendering_sync().data[0]
    assert not out[:, :2 * RQ].any()
    # the buffer source plays noise[50:] from frame 256 on (+ the constant from 677 on)
    np.testing.assert_array_equal(out[0, 2 * RQ:5 * RQ], noise[0, 0, 50:50 + 3 * RQ])
    assert np.allclose(out[0, 5 * RQ + 38:9 * RQ] - noise[0, 0, 50 + 3 * RQ + 38:50 + 7 * RQ], 0.125, atol=1e-6)
    # stopped at quantum 9: from there only the constant and the oscillator
    assert np.abs(out[:, 9 * RQ:14 * RQ]).max() <= 0.125 + 0.2 + 1e-6
    assert not out[:, 17 * RQ:].any() or np.abs(out[:, 17 * RQ:]).max() <= 0.2 + 1e-6
    ctx.close()


@pytest.mark.gpu
def test_sources_started_and_stopped_inside_callbacks_match_the_oracle(hip, orc):
    noise = white_noise(3, 2, RQ * 30)
    outs = []
    for binding in (hip, orc):
        ctx, _ = _sources_at_suspend(binding, noise)
        outs.append(ctx.start_rendering_sync().data)
        ctx.close()
    assert np.abs(outs[0]).max() > 1e-2
    assert rms_err(outs[0], outs[1]).max() <= 1e-6


def _count_sensitive(binding, noise):
    """a stereo branch that joins and leaves a StereoPanner / Delay chain: while it is away the chain's input is silent MONO
    (quantum.rs: an input without connections), so the delay's ring is re-mixed and the panner switches formulas"""
    n_inst = noise.shape[0]
    ctx = waa.OfflineAudioContext(2, RQ * 30, SR, n_instances=n_inst, binding=binding)
    src = ctx.create_buffer_source()
    src.set_buffer_batch(noise, SR)
    mono = ctx.create_constant_source()
    mono.offset.set_value(0.05)
    dly = ctx.create_delay(0.25)
    dly.delay_time.set_value(0.003)
    pan = ctx.create_stereo_panner(pan=0.4)
    flt = ctx.create_biquad_filter(type_="peaking", frequency=2000.0, q=3.0, gain=6.0)
    mono.connect(dly)
    dly.connect(flt).connect(pan).connect(ctx.destination())
    src.start()
    mono.start()
    ctx.suspend_sync(4 * RQ / SR, lambda c: src.connect(dly))
    ctx.suspend_sync(13 * RQ / SR, lambda c: src.disconnect(dly))
    ctx.suspend_sync(21 * RQ / SR, lambda c: src.connect(dly))
    return ctx


@pytest.mark.gpu
def test_channel_counts_follow_connections_made_at_suspend_points(hip, orc):
    noise = white_noise(3, 2, RQ * 30)
    outs = []
    for binding in (hip, orc):
        ctx = _count_sensitive(binding, noise)
        outs.append(ctx.start_rendering_sync().data)
        ctx.close()
    assert np.abs(outs[0]).max() > 1e-2
    assert rms_err(outs[0], outs[1]).max() <= 1e-6


@pytest.mark.gpu
def test_param_connection_made_at_a_suspend_point(hip, orc):
    """node.connect(&param) from a callback: the LFO modulates the gain from quantum 6 on and is cut again at quantum 15"""
    noise = white_noise(2, 2, RQ * 24)
    outs = []
    for binding in (hip, orc):
        ctx = waa.OfflineAudioContext(2, RQ * 24, SR, n_instances=2, binding=binding)
        src = ctx.create_buffer_source()
        src.set_buffer_batch(noise, SR)
        gain = ctx.create_gain(gain=0.5)
        lfo = ctx.create_oscillator(type_="sine", frequency=300.0)
        depth = ctx.create_gain(gain=0.3)
        lfo.connect(depth)
        src.connect(gain).connect(ctx.destination())
        src.start()
        lfo.start()
        ctx.suspend_sync(6 * RQ / SR, lambda c: depth.connect(gain.gain))
        ctx.suspend_sync(15 * RQ / SR, lambda c: depth.disconnect(gain.gain))
        outs.append(ctx.start_rendering_sync().data)
        ctx.close()
    assert rms_err(outs[0], outs[1]).max() <= 1e-6
    # ... and it did something: outside [6, 15) the output is 0.5 x noise
    np.testing.assert_allclose(outs[1][:, :, :6 * RQ], 0.5 * noise[:, :, :6 * RQ], atol=1e-7)
    assert np.abs(outs[1][:, :, 6 * RQ:15 * RQ] - 0.5 * noise[:, :, 6 * RQ:15 * RQ]).max() > 0.05


def _analyser_pull_inside_a_callback(binding, noise):
    """a callback that READS rendered audio: the time-domain data of an analyser at the suspend point decides the gain that follows
    (the product renders node-major: api.py renders the quanta in front of the suspend point with a batch of their own)"""
    ctx = waa.OfflineAudioContext(1, RQ * 20, SR, n_instances=noise.shape[0], binding=binding)
    src = ctx.create_buffer_source()
    src.set_buffer_batch(noise, SR)
    gain = ctx.create_gain(gain=1.0)
    an = ctx.create_analyser(fft_size=256)
    src.connect(gain).connect(ctx.destination())
    gain.connect(an)
    src.start()
    seen = {}

    def q8(c):
        td = an.get_float_time_domain_data_all()
        seen["td"] = td.copy()
        gain.gain.set_value(float(np.float32(0.5 / max(np.abs(td).max(), 1e-3))))

    ctx.suspend_sync(8 * RQ / SR, q8)
    out = ctx.start_rendering_sync().data
    ctx.close()
    return out, seen["td"]


def test_callback_reads_rendered_audio_oracle(orc):
    noise = white_noise(2, 1, RQ * 20)
    out, td = _analyser_pull_inside_a_callback(orc, noise)
    # the pull saw the last 256 frames in front of quantum 8 of every context
    np.testing.assert_array_equal(td, noise[:, 0, 8 * RQ - 256:8 * RQ])
    g = np.float32(0.5 / np.abs(td).max())
    np.testing.assert_array_equal(out[:, 0, :8 * RQ], noise[:, 0, :8 * RQ])
    np.testing.assert_allclose(out[:, 0, 8 * RQ:], noise[:, 0, 8 * RQ:] * g, rtol=1e-6)


@pytest.mark.gpu
def test_callback_reads_rendered_audio_device(hip, orc):
    noise = white_noise(3, 1, RQ * 20)
    got, td_hip = _analyser_pull_inside_a_callback(hip, noise)
    ref, td_orc = _analyser_pull_inside_a_callback(orc, noise)
    np.testing.assert_array_equal(td_hip, td_orc)
    assert rms_err(got, ref).max() <= 1e-6
