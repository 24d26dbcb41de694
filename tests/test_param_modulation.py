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


def test_modulating_a_host_evaluated_param_is_out_of_scope(be):
    c = ctx(be, 1, RQ)
    src = c.create_buffer_source()
    src.set_buffer(waa.AudioBuffer(np.ones((1, RQ), np.float32), 48000.0))
    lfo = c.create_constant_source(offset=0.1)
    lfo.connect(src.playback_rate)
    src.connect(c.destination())
    src.start()
    lfo.start()
    with pytest.raises(waa.WaaError) as ei:
        c.start_rendering_sync()
    assert ei.value.status == 4


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
