"""render_contexts (web-audio-api-rs_amd/mixed.py; round-5 review, missing 4): contexts built one by one — the reference's unit,
src/context/offline.rs:78-143 — of DIFFERENT graph shapes, bucketed by shape, every bucket rendered as one batch, the AudioBuffers
handed back in the callers' order.  Each result equals what the context renders on its own."""
import numpy as np
import pytest

import web_audio_api_rs_amd as waa
from web_audio_api_rs_amd.mixed import bucket_report, render_contexts, shape_key
from graphs import white_noise

RQ = 128
SR = 48000.0
FRAMES = RQ * 30 + 7


def _requests(binding, device=-1):
    """nine "requests": three graph shapes, per-context buffers / parameters / automation / start times"""
    rng = np.random.default_rng(5)
    ctxs = []
    ir = waa.AudioBuffer((rng.standard_normal((2, 300)) * np.exp(-np.arange(300) / 60.0)).astype(np.float32), SR)

    def noise(ch, frames=FRAMES):
        return waa.AudioBuffer(rng.uniform(-1, 1, (ch, frames)).astype(np.float32), SR)

    for k in range(4):  # shape A: source -> Biquad -> Gain -> destination; everything per context
        c = waa.OfflineAudioContext(2, FRAMES, SR, binding=binding, device=device)
        s = c.create_buffer_source()
        s.set_buffer(noise(2))
        f = c.create_biquad_filter(type_="lowpass", frequency=300.0 + 200.0 * k, q=1.0 + 0.3 * k)
        g = c.create_gain(gain=0.5)
        if k % 2:
            g.gain.linear_ramp_to_value_at_time(0.1 + 0.2 * k, 0.05)
        s.connect(f).connect(g).connect(c.destination())
        s.start_at(0.001 * k)
        ctxs.append(c)
    for k in range(3):  # shape B: source -> Convolver (the SAME response object) -> StereoPanner -> destination
        c = waa.OfflineAudioContext(2, FRAMES, SR, binding=binding, device=device)
        s = c.create_buffer_source()
        s.set_buffer(noise(1))
        cv = c.create_convolver(buffer=ir)
        p = c.create_stereo_panner(pan=-0.5 + 0.5 * k)
        s.connect(cv).connect(p).connect(c.destination())
        s.start()
        if k == 2:
            s.set_loop(True)
            s.set_loop_end(0.01)
        ctxs.append(c)
    c = waa.OfflineAudioContext(2, FRAMES, SR, binding=binding, device=device)  # shape A again, but another biquad TYPE: its own bucket
    s = c.create_buffer_source()
    s.set_buffer(noise(2))
    s.connect(c.create_biquad_filter(type_="highpass", frequency=900.0)).connect(c.create_gain(gain=0.5)).connect(c.destination())
    s.start()
    ctxs.append(c)
    c = waa.OfflineAudioContext(1, FRAMES, SR, binding=binding, device=device)  # an oscillator patch, mono context
    o = c.create_oscillator(type_="square", frequency=330.0)
    o.connect(c.destination())
    o.start_at(0.002)
    o.stop_at(0.05)
    ctxs.append(c)
    return ctxs


def test_buckets_follow_the_shape(orc):
    ctxs = _requests(orc)
    assert bucket_report(ctxs) == [[0, 1, 2, 3], [4, 5, 6], [7], [8]]
    assert shape_key(ctxs[0]) == shape_key(ctxs[3]) != shape_key(ctxs[7])
    # a context with a suspend callback renders on its own
    ctxs[1].suspend_sync(0.01, lambda c: None)
    assert bucket_report(ctxs) == [[0, 2, 3], [1], [4, 5, 6], [7], [8]]
    for c in ctxs:
        c.close()


def _check(binding, device=-1, exact=True):
    alone = []
    for c in _requests(binding, device):
        alone.append(c.start_rendering_sync().data)
        c.close()
    together = render_contexts(_requests(binding, device))
    assert len(together) == len(alone)
    for k, (a, t) in enumerate(zip(alone, together)):
        assert t.data.shape == a.shape == (1, a.shape[1], FRAMES)
        assert np.abs(a).max() > 1e-3
        if exact or k not in (4, 5, 6):
            assert np.array_equal(t.data, a), k  # the batch changes nothing: same bits as the context on its own
        else:
            # (the device's FFT convolver packs PAIRS of contexts into one complex transform: a context's roundoff depends on its
            # partner — 1e-9 of full scale, DESIGN.md section 5 — so a context of a batch of three is not bit-identical to itself alone)
            assert np.sqrt(np.mean((t.data.astype(np.float64) - a) ** 2, axis=-1)).max() <= 2e-7, k
    return together


def test_mixed_shapes_oracle(orc):
    _check(orc)


@pytest.mark.gpu
def test_mixed_shapes_device(hip, orc):
    got = _check(hip, 0, exact=False)
    ref = _check(orc)
    for g, r in zip(got, ref):
        assert np.sqrt(np.mean((g.data.astype(np.float64) - r.data) ** 2, axis=-1)).max() <= 1e-6
