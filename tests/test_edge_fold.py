"""Producers folded into the input stage of a summing node: a source whose only consumer sums
several inputs is fetched by the summing kernel itself, and a GainNode between a signal (or a source) and such a
consumer becomes a per-edge gain.  The mixer pattern source -> Gain -> bus then costs no pass through HBM of its
own; the result has to stay what the reference's per-node walk computes (gain.rs:143-199 special cases included)."""
import numpy as np
import pytest

import web_audio_api_rs_amd as waa
from graphs import rms_err, white_noise

RQ = 128
SR = 48000.0
FRAMES = 2048 * 2 + 77


def mixer(be, n_inst=3, device=None):
    kw = {} if device is None else {"device": device}
    c = waa.OfflineAudioContext(2, FRAMES, SR, n_instances=n_inst, binding=be, **kw)
    nq = (FRAMES + RQ - 1) // RQ
    bus = c.destination()
    # 1: stereo source -> constant gain (per instance) -> bus
    s1 = c.create_buffer_source()
    s1.set_buffer_batch(white_noise(n_inst, 2, FRAMES, seed0=1), SR)
    g1 = c.create_gain(gain=0.7)
    for i in range(n_inst):
        g1.gain.set_value(0.3 + 0.2 * i, instance=i)
    s1.connect(g1).connect(bus)
    # 2: mono source (up-mixed by the bus AFTER the gain) -> k-rate gain with exact zeros and ones -> bus
    s2 = c.create_buffer_source()
    s2.set_buffer_batch(white_noise(n_inst, 1, FRAMES, seed0=2), SR)
    g2 = c.create_gain(gain=0.5)
    vals = np.linspace(-1.0, 1.0, nq).astype(np.float32)
    vals[3::5] = 1.0
    vals[4::7] = 0.0
    g2.gain.set_block(0, vals)
    s2.connect(g2).connect(bus)
    # 3: a bare stereo source straight into the bus
    s3 = c.create_buffer_source()
    s3.set_buffer_batch(white_noise(n_inst, 2, FRAMES, seed0=3), SR)
    s3.connect(bus)
    # 4: a filtered (materialised) signal with two consumers, one of them through an a-rate gain
    s4 = c.create_buffer_source()
    s4.set_buffer_batch(white_noise(n_inst, 2, FRAMES, seed0=4), SR)
    bq = c.create_biquad_filter(type_="lowpass", frequency=2000.0)
    g4 = c.create_gain(gain=1.0)
    g4.gain.set_value_at_time(0.0, 0.0).linear_ramp_to_value_at_time(1.0, FRAMES / SR)
    s4.connect(bq)
    bq.connect(g4).connect(bus)
    bq.connect(bus)
    for s in (s1, s2, s3, s4):
        s.start()
    return c


def test_mixer_plan_folds_sources_and_gains(hip):
    c = mixer(hip, device=waa.PLAN_ONLY)
    plan = c.plan_describe()
    assert plan.count("folded into an input edge") == 3
    # five edges into the destination: one partial sum of four + the final sum; apart from the BiquadFilter's output
    # (two consumers) nothing is materialised: no launch that only copies a source or only applies a gain
    lines = plan.splitlines()
    assert sum("fan-in partial sum of 4 inputs" in l for l in lines) == 1
    assert not any(l.startswith("chain ") and ("in=[source:" in l or "ops=[GAIN]" in l) for l in lines)
    assert lines[-1].startswith("chain parallel C=2 in=[signal:2ch+gain*source:2ch]")
    c.close()


@pytest.mark.measure
def test_fold_can_be_disabled(hip, monkeypatch):
    monkeypatch.setenv("WAA_NO_EDGE_FOLD", "1")
    c = mixer(hip, device=waa.PLAN_ONLY)
    assert "folded" not in c.plan_describe()
    c.close()


@pytest.mark.gpu
def test_mixer_parity(hip, orc):
    g = mixer(hip).start_rendering_sync().data
    o = mixer(orc).start_rendering_sync().data
    assert rms_err(g, o).max() <= 1e-7
    assert np.abs(g - o).max() <= 2e-6


@pytest.mark.measure
@pytest.mark.gpu
def test_fold_matches_unfolded(hip, monkeypatch):
    """the folded plan and the node-per-launch plan are the same arithmetic in the same order: bit-identical"""
    a = mixer(hip).start_rendering_sync().data
    monkeypatch.setenv("WAA_NO_EDGE_FOLD", "1")
    b = mixer(hip).start_rendering_sync().data
    assert np.array_equal(a, b)
