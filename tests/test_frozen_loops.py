"""Frozen-state nodes (WaveShaper 2x / 4x, HRTF PannerNode) INSIDE feedback loops (round 5; status 4 until round 4).  The
reference renders any node in a cycle that holds a DelayNode (graph.rs:331-487, delay.rs:693-701).  On the device the loop is
cut at the frozen-state node(s): [the items in front -> the node's mixed input] [its link / transform / FIR launches] [the items
behind], every launch over the same render quantum, quantum after quantum (Step::qgroup; ranged dyn_kernel launches that keep
the items' state in memory, a DelayNode split over two launches talking through memory).  Against the oracle, which restates
the reference per quantum."""
import os

import numpy as np
import pytest

import web_audio_api_rs_amd as waa
from graphs import rms_err, white_noise

pytestmark = pytest.mark.gpu
RQ, SR, N = 128, 48000.0, 3
FRAMES = RQ * 70 + 33
CURVE = np.tanh(np.linspace(-2.5, 2.5, 129)).astype(np.float32)
OFFSET_CURVE = (np.tanh(np.linspace(-2.0, 2.0, 64)) + 0.25).astype(np.float32)   # maps 0 to a signal: never skips a block


def _frozen(c, kind):
    if kind in ("2x", "4x"):
        return c.create_wave_shaper(curve=CURVE, oversample=kind)
    if kind == "2x-offset":
        return c.create_wave_shaper(curve=OFFSET_CURVE, oversample="2x")
    return c.create_panner(panning_model="HRTF", position=(0.8, 0.3, -0.6))


def _compare(build, hip, orc, what="frozen-state node inside"):
    outs = []
    for be in (hip, orc):
        c = build(be)
        if be is hip:
            plan = c.plan_describe()
            assert what in plan, plan
        outs.append(c.start_rendering_sync().data)
        c.close()
    g, o = outs
    assert np.isfinite(g).all() and np.isfinite(o).all()
    assert float(np.abs(o).max()) > 1e-3
    scale = max(1.0, float(np.abs(o).max()))
    assert rms_err(g, o).max() <= 1e-6 * scale, (rms_err(g, o), scale)
    assert np.abs(g - o).max() <= 2e-5 * scale, float(np.abs(g - o).max())
    return g


@pytest.mark.parametrize("delay_time", [0.0, 0.0007, 0.01, 0.06])
@pytest.mark.parametrize("kind", ["2x", "4x", "hrtf", "2x-offset"])
def test_echo_with_a_frozen_state_node_in_the_loop(hip, orc, kind, delay_time):
    """source (a burst that ends: the loop rings on, falls silent, the node's state freezes) -> Delay -> node -> Gain -> back"""
    def build(be):
        c = waa.OfflineAudioContext(2, FRAMES, SR, n_instances=N, binding=be)
        src = c.create_buffer_source()
        src.set_buffer_batch(white_noise(N, 2, RQ * 20 + 7, seed0=5) * 0.6, SR)
        d = c.create_delay(0.1, delay_time=delay_time)
        f = _frozen(c, kind)
        g = c.create_gain(gain=0.45)
        src.connect(d)
        d.connect(f).connect(g).connect(d)
        f.connect(c.destination())
        for i in range(N):
            src.start_at(i * 211.0 / SR, instance=i)
        return c
    _compare(build, hip, orc)


def test_two_frozen_state_nodes_and_a_filter_in_one_loop(hip, orc):
    """Delay -> WaveShaper 2x -> Biquad -> HRTF panner -> Gain -> back: three segments, the filter's state travels through memory"""
    def build(be):
        c = waa.OfflineAudioContext(2, FRAMES, SR, n_instances=N, binding=be)
        a = c.create_buffer_source()
        a.set_buffer_batch(white_noise(N, 1, RQ * 30, seed0=8) * 0.5, SR)
        b2 = c.create_buffer_source()
        b2.set_buffer_batch(white_noise(N, 2, RQ * 12, seed0=9) * 0.5, SR)
        d = c.create_delay(0.2, delay_time=0.004)
        sh = c.create_wave_shaper(curve=CURVE, oversample="2x")
        bq = c.create_biquad_filter(type_="lowpass", frequency=2500.0, q=2.0)
        pn = c.create_panner(panning_model="HRTF", position=(-1.0, 0.0, 0.4))
        g = c.create_gain(gain=0.3)
        a.connect(d)
        b2.connect(d)
        d.connect(sh).connect(bq).connect(pn).connect(g).connect(d)
        pn.connect(c.destination())
        a.start()
        b2.start_at(0.02)
        return c
    _compare(build, hip, orc)


def test_delay_pair_that_stays_inside_one_segment(hip, orc):
    """the loop's breaking delay sits behind the node (writer and reader in one launch), a second plain delay in front of it"""
    def build(be):
        c = waa.OfflineAudioContext(2, FRAMES, SR, n_instances=N, binding=be)
        src = c.create_buffer_source()
        src.set_buffer_batch(white_noise(N, 2, RQ * 25, seed0=12) * 0.5, SR)
        mix = c.create_gain(gain=1.0)
        sh = c.create_wave_shaper(curve=CURVE, oversample="4x")
        g = c.create_gain(gain=0.5)
        d = c.create_delay(0.1, delay_time=0.02)
        src.connect(mix)
        mix.connect(sh).connect(g).connect(d).connect(mix)
        sh.connect(c.destination())
        src.start()
        return c
    _compare(build, hip, orc)


@pytest.mark.measure
def test_the_old_refusal_is_still_there_behind_its_switch(hip, monkeypatch):
    monkeypatch.setenv("WAA_NO_FROZEN_LOOPS", "1")
    c = waa.OfflineAudioContext(2, FRAMES, SR, n_instances=1, binding=hip)
    src = c.create_buffer_source()
    src.set_buffer_batch(white_noise(1, 2, RQ * 4), SR)
    d = c.create_delay(0.1, delay_time=0.01)
    f = _frozen(c, "2x")
    src.connect(d)
    d.connect(f).connect(c.create_gain(gain=0.4)).connect(d)
    f.connect(c.destination())
    src.start()
    with pytest.raises(waa.WaaError) as e:
        c.plan_describe()
    assert e.value.status == 4
    c.close()


def test_automated_and_modulated_params_inside_such_a_loop(hip, orc):
    """a chorus-like sweep of delayTime (a-rate), an automated Biquad, a gain modulated by an oscillator OUTSIDE the loop: their
    tables and summing chains are launched once in front of the blocks (fuzz seeds 100961 / 101141 / 103054 / 103251 of r05h:
    left between the loop's launches they cut the quantum-blocked loop in two)"""
    def build(be):
        c = waa.OfflineAudioContext(2, FRAMES, SR, n_instances=N, binding=be)
        src = c.create_buffer_source()
        src.set_buffer_batch(white_noise(N, 2, RQ * 40, seed0=21) * 0.5, SR)
        d = c.create_delay(0.05, delay_time=0.01)
        d.delay_time.set_value_at_time(0.004, 0.0).linear_ramp_to_value_at_time(0.03, FRAMES / SR)
        sh = c.create_wave_shaper(curve=CURVE, oversample="2x")
        bq = c.create_biquad_filter(type_="bandpass", frequency=800.0, q=1.5)
        bq.frequency.set_value_at_time(300.0, 0.0).exponential_ramp_to_value_at_time(4000.0, FRAMES / SR)
        g = c.create_gain(gain=0.35)
        lfo = c.create_oscillator(type_="sine", frequency=7.0)
        lfo.connect(c.create_gain(gain=0.1)).connect(g.gain)
        lfo.start()
        src.connect(d)
        d.connect(sh).connect(bq).connect(g).connect(d)
        bq.connect(c.destination())
        src.start()
        return c
    _compare(build, hip, orc)


def _ir(n_ch, taps, seed):
    rng = np.random.default_rng(seed)
    return (rng.uniform(-1, 1, (n_ch, taps)) * np.exp(-np.arange(taps) / (0.3 * taps))[None, :]).astype(np.float32)


@pytest.mark.parametrize("delay_time", [0.0, 0.003, 0.01, 0.06])
@pytest.mark.parametrize("ir_ch,src_ch,taps", [(2, 2, 700), (1, 1, 100), (2, 1, 16), (2, 2, 3000)])
def test_echo_with_a_short_convolver_in_the_loop(hip, orc, ir_ch, src_ch, taps, delay_time):
    """a ConvolverNode whose response has 128-frame partitions (at most 24 x 128 taps) inside a loop of any delay: its transforms
    follow the loop quantum by quantum like the frozen-state nodes (status 4 until round 4 for delays below a partition / dynamic counts)"""
    def build(be):
        c = waa.OfflineAudioContext(2, FRAMES, SR, n_instances=N, binding=be)
        src = c.create_buffer_source()
        src.set_buffer_batch(white_noise(N, src_ch, RQ * 22 + 5, seed0=31) * 0.5, SR)
        d = c.create_delay(0.1, delay_time=delay_time)
        cv = c.create_convolver(buffer=waa.AudioBuffer(_ir(ir_ch, taps, taps), SR))
        g = c.create_gain(gain=0.3)
        src.connect(d)
        d.connect(cv).connect(g).connect(d)
        cv.connect(c.destination())
        for i in range(N):
            src.start_at(i * 97.0 / SR, instance=i)
        return c
    if delay_time >= 0.06 and taps <= 700:
        # a delay longer than a tile and static counts: may be the block-scheduled static loop of round 3 instead
        _compare(build, hip, orc, what="convolver node")
    else:
        _compare(build, hip, orc)


def test_convolver_and_oversampled_shaper_in_one_loop(hip, orc):
    def build(be):
        c = waa.OfflineAudioContext(2, FRAMES, SR, n_instances=N, binding=be)
        src = c.create_buffer_source()
        src.set_buffer_batch(white_noise(N, 2, RQ * 30, seed0=33) * 0.4, SR)
        d = c.create_delay(0.1, delay_time=0.005)
        cv = c.create_convolver(buffer=waa.AudioBuffer(_ir(2, 300, 7), SR))
        sh = c.create_wave_shaper(curve=CURVE, oversample="2x")
        g = c.create_gain(gain=0.25)
        src.connect(d)
        d.connect(cv).connect(sh).connect(g).connect(d)
        sh.connect(c.destination())
        src.start()
        return c
    _compare(build, hip, orc)


def test_what_stays_out_of_scope_in_a_loop(hip):
    """a response longer than 24 x 128 taps in a short loop (its partitions span several quanta), and a mono response behind a
    stereo input (channel 1 runs in compacted time): status 4"""
    for ir, src_ch in ((_ir(2, 5000, 1), 2), (_ir(1, 100, 2), 2)):
        c = waa.OfflineAudioContext(2, FRAMES, SR, n_instances=1, binding=hip)
        src = c.create_buffer_source()
        src.set_buffer_batch(white_noise(1, src_ch, RQ * 9 + 3), SR)
        d = c.create_delay(0.1, delay_time=0.003)
        cv = c.create_convolver(buffer=waa.AudioBuffer(ir, SR))
        src.connect(d)
        d.connect(cv).connect(c.create_gain(gain=0.3)).connect(d)
        cv.connect(c.destination())
        src.start()
        with pytest.raises(waa.WaaError) as e:
            c.plan_describe()
        assert e.value.status == 4, str(e.value)
        c.close()
