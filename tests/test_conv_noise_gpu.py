"""Silence decisions behind a ConvolverNode (DESIGN 5, 2b — round 5): the reference's FFT convolver leaves roundoff noise, never
exact zeros, for up to two 1024-frame blocks around anything non-zero in its input (the automaton: tests/test_conv_noise.py), and a
DelayNode — silent only when it "read nothing but zeros" (delay.rs:660-668) — stays active on that noise.  The device's overlap-save
transforms put out exact zeros as soon as their window holds nothing else, so without the noise floor (waa_dyn.hip:
conv_floor_kernel) a DelayNode behind a convolver falls silent up to 14 quanta early, the ring of the NEXT DelayNode is re-mixed to
mono (delay.rs:428-489) that much earlier, and whatever stereo signal still sits in that ring comes out as the average of its
channels: a difference of the size of the signal.  Each test builds that situation, or replays a fuzz seed that found it."""
import os

import numpy as np
import pytest

import web_audio_api_rs_amd as waa
from graphs import rms_err, white_noise
from test_fuzz_graphs import build_random_graph

pytestmark = pytest.mark.gpu
RQ = 128
SR = 48000.0
N = 3


def _compare(build, hip, orc, expect_note=True):
    c = build(hip)
    plan = c.plan_describe()
    assert "dynamic-count group" in plan, plan
    g = c.start_rendering_sync().data
    c.close()
    c = build(orc)
    o = c.start_rendering_sync().data
    c.close()
    assert np.isfinite(o).all() and np.isfinite(g).all()
    scale = max(1.0, float(np.abs(o).max()))
    assert rms_err(g, o).max() <= 1e-6 * scale, rms_err(g, o)
    assert np.abs(g - o).max() <= 2e-5 * scale, np.abs(g - o).max()
    return g, o


def _padded(x, quanta=30):
    """the burst followed by digital silence: the source stays active, the convolver keeps being called — on zeros"""
    return np.concatenate([x, np.zeros(x.shape[:2] + (quanta * RQ,), np.float32)], axis=2)


def _burst_graph(be, frames, burst_quanta, start_frames, d1_quanta, d2_quanta, ir, ir_nch=2, pad_quanta=30):
    """stereo burst (R = -L) -> convolver -> DelayNode D1 -> DelayNode D2 -> destination.  D1's silence follows the convolver's
    noise; when it falls silent D2's ring collapses to mono and a burst still inside it is averaged away (R = -L: to nothing)."""
    c = waa.OfflineAudioContext(2, frames, SR, n_instances=N, binding=be)
    x = _padded(white_noise(N, 1, burst_quanta * RQ, seed0=31) * 0.5, pad_quanta)  # (ends mid-render: a dynamic plan)
    src = c.create_buffer_source()
    src.set_buffer_batch(np.concatenate([x, -x], axis=1), SR)
    for i in range(N):
        src.start_at(start_frames[i] / SR, instance=i)
    conv = c.create_convolver(buffer=waa.AudioBuffer(ir[:ir_nch], SR), disable_normalization=True)
    d1 = c.create_delay(1.0, delay_time=d1_quanta * RQ / SR)
    d2 = c.create_delay(1.0, delay_time=d2_quanta * RQ / SR)
    src.connect(conv)
    conv.connect(d1)
    d1.connect(d2)
    d2.connect(c.destination())
    return c


def _short_ir(taps=16, seed=5):
    rng = np.random.default_rng(seed)
    h = np.repeat(rng.standard_normal((1, taps)).astype(np.float32) * 0.3, 2, axis=0)  # (both channels alike: R stays -L)
    h[:, 0] = 1.0
    return h


@pytest.mark.parametrize("d2_quanta", [6, 8, 11])
def test_burst_leaves_the_second_delay_in_stereo(hip, orc, d2_quanta):
    """a two-quantum burst at the start of a block: the reference's convolver is noise to the end of the NEXT block (quantum 15),
    D1 active until then, and the burst leaves D2 (6 … 11 quanta later) in stereo — R = -L, so the mono average would be silence"""
    build = lambda be: _burst_graph(be, RQ * 60, 2, [0, 37, RQ + 5], 3, d2_quanta, _short_ir())
    g, o = _compare(build, hip, orc)
    assert np.abs(o).max(axis=(1, 2)).min() > 0.05 and np.abs(o[:, 0] + o[:, 1]).max() < 1e-3  # every context: the burst, in stereo


@pytest.mark.measure
def test_burst_without_the_floor_is_averaged_away(hip, orc, monkeypatch):
    """the same graph with the floor switched off (measure build): the device's D1 falls silent at quantum 7, D2's ring is mono by the
    time the burst leaves it — the test above does discriminate"""
    monkeypatch.setenv("WAA_NO_CONV_NOISE_FLOOR", "1")
    build = lambda be: _burst_graph(be, RQ * 60, 2, [0, 37, RQ + 5], 3, 8, _short_ir())
    c = build(hip)
    g = c.start_rendering_sync().data
    c.close()
    c = build(orc)
    o = c.start_rendering_sync().data
    c.close()
    assert np.abs(g - o).max() > 0.05


@pytest.mark.parametrize("taps,burst_quanta,start", [(16, 1, 187), (1500, 3, 5 * RQ + 64), (3000, 2, 11 * RQ), (5000, 9, 200)])
def test_noise_of_longer_responses(hip, orc, taps, burst_quanta, start):
    """responses of two … five 1024-frame segments (512- and 2048-frame partitions on the device): the noise lasts one block beyond
    the last segment's; D2's delay is swept over that edge by the three contexts' start times"""
    rng = np.random.default_rng(taps)
    h = (rng.standard_normal((2, taps)) * np.exp(-np.arange(taps) / (taps / 3.0))).astype(np.float32) * 0.2
    segs = (taps + 1023) // 1024
    edge = ((start // RQ + burst_quanta) // 8 + segs + 1) * 8  # first quantum of exact zeros in the reference (about)
    d2 = max(2, edge - (start // RQ) - 6)
    build = lambda be: _burst_graph(be, RQ * (edge + d2 + 40), burst_quanta, [start, start + 3 * RQ, start + 6 * RQ + 17], 2, d2, h,
                                   pad_quanta=edge + 10)
    _compare(build, hip, orc)


def test_mono_response_and_mono_input(hip, orc):
    """one FFTConvolver (mono in, mono response): the output count is 1 and only channel 0 is floored"""
    def build(be):
        c = waa.OfflineAudioContext(2, RQ * 60, SR, n_instances=N, binding=be)
        src = c.create_buffer_source()
        src.set_buffer_batch(_padded(white_noise(N, 1, 2 * RQ, seed0=41) * 0.5), SR)
        for i in range(N):
            src.start_at((i * 450) / SR, instance=i)
        conv = c.create_convolver(buffer=waa.AudioBuffer(_short_ir()[:1], SR), disable_normalization=True)
        d1 = c.create_delay(1.0, delay_time=3 * RQ / SR)
        pan = c.create_stereo_panner(pan=0.4)  # (mono / stereo law and a stereo ring behind it)
        other = c.create_buffer_source()
        y = white_noise(N, 1, 2 * RQ, seed0=42) * 0.5
        other.set_buffer_batch(np.concatenate([y, -y], axis=1), SR)
        other.start_at(2 * RQ / SR)
        d2 = c.create_delay(1.0, delay_time=9 * RQ / SR)
        src.connect(conv)
        conv.connect(d1)
        d1.connect(pan)
        pan.connect(d2)
        other.connect(d2)
        d2.connect(c.destination())
        return c
    _compare(build, hip, orc)


def test_true_stereo_response(hip, orc):
    """four channels: four FFTConvolvers, output channel c = convolver c of the left + convolver 2 + c of the right input"""
    rng = np.random.default_rng(8)
    h = rng.standard_normal((4, 40)).astype(np.float32) * 0.2
    def build(be):
        c = waa.OfflineAudioContext(2, RQ * 60, SR, n_instances=N, binding=be)
        x = _padded(white_noise(N, 1, 2 * RQ, seed0=51) * 0.5)
        src = c.create_buffer_source()
        src.set_buffer_batch(np.concatenate([x, -x], axis=1), SR)
        for i in range(N):
            src.start_at((i * 200) / SR, instance=i)
        conv = c.create_convolver(buffer=waa.AudioBuffer(h, SR), disable_normalization=True)
        d1 = c.create_delay(1.0, delay_time=3 * RQ / SR)
        d2 = c.create_delay(1.0, delay_time=8 * RQ / SR)
        thru = c.create_delay(1.0, delay_time=1 * RQ / SR)   # the burst itself reaches D2 as well (stereo, R = -L)
        src.connect(conv)
        conv.connect(d1)
        d1.connect(d2)
        src.connect(thru)
        thru.connect(d2)
        d2.connect(c.destination())
        return c
    _compare(build, hip, orc)


def test_echo_between_two_bursts_keeps_its_ring_in_stereo(hip, orc):
    """fuzz seed 651276's mechanism: a DelayNode in a feedback loop behind a short convolver; between two echoes the loop carries
    nothing but the convolver's noise — in the reference the reader stays active on it and the ring stays stereo"""
    def build(be):
        c = waa.OfflineAudioContext(2, RQ * 120, SR, n_instances=N, binding=be)
        x = _padded(white_noise(N, 1, 3 * RQ, seed0=61) * 0.5, 30)
        src = c.create_buffer_source()
        src.set_buffer_batch(np.concatenate([x, -x], axis=1), SR)
        for i in range(N):
            src.start_at((64 + i * 333) / SR, instance=i)
        conv = c.create_convolver(buffer=waa.AudioBuffer(_short_ir(), SR), disable_normalization=True)
        d = c.create_delay(1.0, delay_time=12.5 * RQ / SR)
        fb = c.create_gain(gain=0.37)
        src.connect(conv)
        conv.connect(d)
        d.connect(fb)
        fb.connect(d)
        d.connect(c.destination())
        return c
    g, o = _compare(build, hip, orc)
    assert np.abs(o[:, 0] + o[:, 1]).max() < 1e-3 and np.abs(o[:, :, 40 * RQ:]).max() > 1e-3  # later echoes are still R = -L


# fuzz seeds of rounds 3 - 5 whose only difference was this class (profiles/r03_fuzz_b.json, r04z_fuzz_*.json, r05m_fuzz.json,
# r05n_fuzz_*.json); (generator, seed) — the generator's draws have changed since round 3, old seeds may build other graphs now
SEEDS = [(False, 403213), (False, 641201), (True, 651276), (True, 115791), (True, 303213), (False, 42294), (False, 1340), (False, 1658)]


@pytest.mark.parametrize("frozen,seed", SEEDS)
def test_fuzz_seeds_of_this_class(hip, orc, frozen, seed):
    ch, descr = build_random_graph(hip, seed, frozen=frozen)
    try:
        g = ch.start_rendering_sync().data
    except waa.WaaError as e:
        if e.status == 4:
            pytest.skip(f"out of scope on the device path: {e} [{descr}]")
        raise
    ch.close()
    co, _ = build_random_graph(orc, seed, frozen=frozen)
    o = co.start_rendering_sync().data
    co.close()
    scale = max(1.0, float(np.abs(o).max()))
    assert rms_err(g, o).max() <= 1e-6 * scale, f"{descr}: rms {rms_err(g, o).max():.3g}"
    assert np.abs(g - o).max() <= 2e-5 * scale, f"{descr}: max |d| {np.abs(g - o).max():.3g}"


@pytest.mark.xfail(reason="DESIGN 5 (2b), last paragraph: a source that PLAYS digital silence counts as active in the planner's replay; "
                          "the plan stays static and nothing tests the DelayNode's data", strict=False)
def test_digital_silence_from_an_active_source_is_not_followed(hip, orc):
    """the known gap next to this class, kept visible: the buffer goes on with zeros to the END of the render (no count change
    anywhere: a static plan); in the reference D1 reads nothing but zeros from quantum 6 on and is silent, D2's ring collapses to
    mono and the burst (R = -L) that leaves it 20 quanta later is averaged away — on the device it comes out in stereo"""
    def build(be):
        c = waa.OfflineAudioContext(2, RQ * 60, SR, n_instances=N, binding=be)
        x = _padded(white_noise(N, 1, 2 * RQ, seed0=71) * 0.5, 60)
        src = c.create_buffer_source()
        src.set_buffer_batch(np.concatenate([x, -x], axis=1), SR)
        src.start_at(0.0)
        d1 = c.create_delay(1.0, delay_time=3 * RQ / SR)
        d2 = c.create_delay(1.0, delay_time=20 * RQ / SR)
        src.connect(d1)
        d1.connect(d2)
        d2.connect(c.destination())
        return c
    c = build(hip)
    assert "dynamic-count group" not in c.plan_describe()
    g = c.start_rendering_sync().data
    c.close()
    c = build(orc)
    o = c.start_rendering_sync().data
    c.close()
    assert np.abs(o).max() == 0.0  # (the reference: averaged away)
    assert np.abs(g - o).max() <= 2e-5
