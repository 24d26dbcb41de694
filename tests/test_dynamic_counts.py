"""The reference's DYNAMIC channel counts, rendered on the device (waa_dyn.hip) and compared with the oracle.

An AudioRenderQuantum carries its own channel count and a silent flag (quantum.rs:109-120, 207-259); a silent quantum is
mono, `add` mixes both operands to the count computed from the receiver's channel config and the current operand counts
(quantum.rs:532-569), and count-sensitive processors follow it.  Each test below builds one situation in which the
count of a signal changes mid-render at such a processor, checks that the planner answers with a dynamic-count group, and
requires the north-star tolerance against the oracle, which restates the reference per quantum.
"""
import numpy as np
import pytest

import web_audio_api_rs_amd as waa
from graphs import rms_err, white_noise

pytestmark = pytest.mark.gpu
RQ = 128
SR = 48000.0
FRAMES = RQ * 90 + 50
N = 3


def _render(build, hip, orc, expect_dynamic=True):
    outs = []
    for be in (hip, orc):
        c = build(be)
        if be is hip:
            plan = c.plan_describe()
            assert ("dynamic-count group" in plan) == expect_dynamic, plan
        outs.append(c.start_rendering_sync().data)
        c.close()
    g, o = outs
    assert np.isfinite(o).all()
    scale = max(1.0, float(np.abs(o).max()))
    assert rms_err(g, o).max() <= 1e-6 * scale, rms_err(g, o)
    assert np.abs(g - o).max() <= 2e-5 * scale
    return g, o


def _ctx(be):
    return waa.OfflineAudioContext(2, FRAMES, SR, n_instances=N, binding=be)


def _buf(c, nch, frames, start=0.0, seed=1, per_instance_start=False):
    s = c.create_buffer_source()
    s.set_buffer_batch(white_noise(N, nch, frames, seed0=seed) * 0.5, SR)
    if per_instance_start:
        for i in range(N):
            s.start_at(start + i * 300.0 / SR, instance=i)
    else:
        s.start_at(start)
    return s


@pytest.mark.parametrize("kind", ["biquad", "iir"])
def test_filter_starts_its_second_channel_from_zero(hip, orc, kind):
    """biquad_filter.rs:798-815 / iir_filter.rs:336-360: mono from t = 0, a stereo source joins later (per instance at a
    different, sub-quantum time) and ends early; the filter rings on in stereo until its state is denormal."""
    def build(be):
        c = _ctx(be)
        mono = _buf(c, 1, FRAMES, seed=11)
        stereo = _buf(c, 2, RQ * 30, start=RQ * 9.5 / SR, seed=12, per_instance_start=True)
        if kind == "biquad":
            f = c.create_biquad_filter(type_="bandpass", frequency=900.0, q=3.0)
        else:
            from scipy import signal
            b, a = signal.butter(3, 0.2)
            f = c.create_iir_filter(b, a)
        mono.connect(f)
        stereo.connect(f)
        f.connect(c.destination())
        return c
    g, o = _render(build, hip, orc)
    assert np.abs(o[:, 0] - o[:, 1]).max() > 1e-3  # (the channels really differ once the stereo source plays)


def test_stereo_panner_switches_between_its_mono_and_stereo_law(hip, orc):
    """stereo_panner.rs:218-317: mono law while only the oscillator plays, stereo law while the buffer plays too, and
    silence (a mono quantum) while a k-rate gain in front of it is zero (gain.rs:163-170)."""
    def build(be):
        c = _ctx(be)
        osc = c.create_oscillator(type_="sawtooth", frequency=330.0)
        osc.start_at(RQ * 2.25 / SR)
        stereo = _buf(c, 2, RQ * 40, start=RQ * 20 / SR, seed=21)
        g = c.create_gain(gain=1.0)
        vals = np.ones((FRAMES + RQ - 1) // RQ, np.float32)
        vals[50:60] = 0.0
        g.gain.set_block(0, vals)
        pan = c.create_stereo_panner(pan=-0.4)
        pan.pan.set_value_at_time(-0.9, 0.0).linear_ramp_to_value_at_time(0.8, FRAMES / SR)
        osc.connect(g)
        stereo.connect(g)
        g.connect(pan).connect(c.destination())
        return c
    _render(build, hip, orc)


def test_equal_power_panner_mono_and_stereo_law(hip, orc):
    """panner.rs:988-1057 with per-instance positions; the stereo source starts late and stops before the end."""
    def build(be):
        c = _ctx(be)
        mono = _buf(c, 1, FRAMES, seed=31)
        stereo = _buf(c, 2, RQ * 35, start=RQ * 12 / SR, seed=32)
        pn = c.create_panner(distance_model="inverse", ref_distance=1.0)
        for i in range(N):
            pn.position_x.set_value(-1.5 + 1.5 * i, instance=i)
            pn.position_z.set_value(-1.0, instance=i)
        mono.connect(pn)
        stereo.connect(pn)
        pn.connect(c.destination())
        return c
    _render(build, hip, orc)


@pytest.mark.parametrize("delay_time", [0.0007, 0.02])
def test_delay_line_is_remixed_to_the_count_of_its_input(hip, orc, delay_time):
    """delay.rs:469-489: a stereo burst, then only a mono source: the line collapses to mono (0.5 (L + R)) the moment
    the input narrows, a second stereo burst widens it again by copying; the reader reports silence from the data."""
    def build(be):
        c = _ctx(be)
        burst1 = _buf(c, 2, RQ * 6, start=RQ * 3 / SR, seed=41)
        burst2 = _buf(c, 2, RQ * 5, start=RQ * 40 / SR, seed=42)
        mono = _buf(c, 1, RQ * 30, start=RQ * 8.5 / SR, seed=43)
        d = c.create_delay(0.1, delay_time=delay_time)
        for s in (burst1, burst2, mono):
            s.connect(d)
        d.connect(c.destination())
        return c
    _render(build, hip, orc)


def test_feedback_loop_with_a_late_stereo_source(hip, orc):
    """a mono oscillator feeds a Delay <-> Biquad -> Gain loop from t = 0, a stereo source joins the loop later: the
    in-loop reader sees the line's count of the PREVIOUS quantum (delay.rs:535-541), the filter starts channel 1 late."""
    def build(be):
        c = _ctx(be)
        osc = c.create_oscillator(frequency=180.0)
        osc.start()
        stereo = _buf(c, 2, RQ * 25, start=RQ * 15 / SR, seed=51)
        d = c.create_delay(0.05, delay_time=0.004)
        bq = c.create_biquad_filter(type_="lowpass", frequency=2500.0)
        fb = c.create_gain(gain=0.6)
        osc.connect(d)
        stereo.connect(d)
        d.connect(bq).connect(fb).connect(d)
        bq.connect(c.destination())
        return c
    _render(build, hip, orc)


@pytest.mark.parametrize("ir_ch", [1, 2])
def test_mono_ir_convolver_freezes_its_second_convolver(hip, orc, ir_ch):
    """convolver.rs:384-407: with a mono impulse response the second FFTConvolver only runs on stereo quanta.  Stereo
    burst, silence (the tail comes out of convolver 0 alone, in both channels), a mono source, and a SECOND stereo burst
    that resumes convolver 1 with whatever state it was frozen in — rendered in compacted time on the device."""
    def build(be):
        c = _ctx(be)
        burst1 = _buf(c, 2, RQ * 6, start=RQ * 2 / SR, seed=61)
        mono = _buf(c, 1, RQ * 10, start=RQ * 20 / SR, seed=62)
        burst2 = _buf(c, 2, RQ * 8, start=RQ * 26 / SR, seed=63)
        rng = np.random.default_rng(64)
        ir = (rng.uniform(-1, 1, (ir_ch, 1700)) * np.exp(-np.arange(1700) / 500.0)).astype(np.float32)
        conv = c.create_convolver(buffer=waa.AudioBuffer(ir, SR))
        for s in (burst1, mono, burst2):
            s.connect(conv)
        conv.connect(c.destination())
        return c
    _render(build, hip, orc)


def test_waveshaper_that_maps_silence_to_a_signal(hip, orc):
    """waveshaper.rs:498-509: a curve whose centre is not 0 turns a silent (mono) input into a constant mono signal; the
    StereoPanner behind it then runs its mono law until the stereo source starts."""
    def build(be):
        c = _ctx(be)
        stereo = _buf(c, 2, RQ * 30, start=RQ * 10 / SR, seed=71)
        sh = c.create_wave_shaper(curve=np.array([0.2, 0.3, 0.5, 0.7, 0.9], np.float32))
        pan = c.create_stereo_panner(pan=0.25)
        stereo.connect(sh).connect(pan).connect(c.destination())
        return c
    _render(build, hip, orc)


def test_loop_members_outside_the_static_loop_kernel_use_the_dynamic_path(hip, orc):
    """An IIRFilter and an a-rate StereoPanner inside a short feedback loop: the static loop kernel does not cover them;
    the planner renders the graph with dyn_kernel instead of refusing it (status 4 in round 1)."""
    def build(be):
        from scipy import signal
        c = _ctx(be)
        src = _buf(c, 2, FRAMES, seed=81)
        d = c.create_delay(0.05, delay_time=0.003)
        b, a = signal.butter(2, 0.3)
        f = c.create_iir_filter(b, a)
        pan = c.create_stereo_panner(pan=0.0)
        pan.pan.set_value_at_time(-1.0, 0.0).linear_ramp_to_value_at_time(1.0, FRAMES / SR)
        fb = c.create_gain(gain=0.5)
        src.connect(d)
        d.connect(f).connect(pan).connect(fb).connect(d)
        pan.connect(c.destination())
        return c
    _render(build, hip, orc)


@pytest.mark.parametrize("wide,interp", [(4, "speakers"), (6, "speakers"), (4, "discrete")])
def test_layouts_above_stereo_change_their_count_mid_render(hip, orc, wide, interp):
    """round 3 (dyn_kernel<6>): a mono source from t = 0, a quad / 5.1 source that joins later and ends early, a stereo one on top —
    through a Gain with an explicit wide channel count, a Biquad, an IIR filter and a WaveShaper (per-channel state for every
    channel that appears, quantum.rs' up- and down-mix tables in both directions), down to the stereo destination"""
    def build(be):
        c = _ctx(be)
        mono = _buf(c, 1, FRAMES, seed=21)
        big = _buf(c, wide, RQ * 25, start=RQ * 7.25 / SR, seed=22, per_instance_start=True)
        stereo = _buf(c, 2, RQ * 40, start=RQ * 20.0 / SR, seed=23)
        bus = c.create_gain(gain=0.7, channel_count=wide, channel_count_mode="max", channel_interpretation=interp)
        from scipy import signal
        b, a = signal.butter(2, 0.3)
        f1 = c.create_biquad_filter(type_="peaking", frequency=1500.0, q=2.0, gain=6.0)
        f2 = c.create_iir_filter(b, a)
        sh = c.create_wave_shaper(curve=np.tanh(np.linspace(-2.0, 2.0, 65)).astype(np.float32))
        for s in (mono, big, stereo):
            s.connect(bus)
        bus.connect(f1).connect(f2).connect(sh).connect(c.destination())
        f1.connect(c.destination())
        return c
    g, o = _render(build, hip, orc)
    assert np.abs(o).max() > 0.1


@pytest.mark.parametrize("delay_time,feedback", [(0.0007, False), (0.02, False), (0.02, True)])
def test_delay_line_above_stereo_is_remixed_in_place(hip, orc, delay_time, feedback):
    """delay.rs:469-489 with layouts above stereo: mono -> quad -> stereo -> mono inputs; every stored quantum of the ring goes
    through `mix(new count, Speakers)` at each change (4 -> 2 is a computed down-mix, 2 -> 4 pads: the chain has no closed
    form, the kernel re-mixes the line in place); also with the delay inside a feedback loop"""
    def build(be):
        c = _ctx(be)
        mono = _buf(c, 1, FRAMES, seed=31)
        quad = _buf(c, 4, RQ * 18, start=RQ * 6.5 / SR, seed=32, per_instance_start=True)
        stereo = _buf(c, 2, RQ * 50, start=RQ * 15.0 / SR, seed=33)
        d = c.create_delay(0.1, delay_time=delay_time)
        for s in (mono, quad, stereo):
            s.connect(d)
        tail = d.connect(c.create_biquad_filter(type_="lowpass", frequency=3000.0))
        if feedback:
            tail.connect(c.create_gain(gain=0.4)).connect(d)
        tail.connect(c.destination())
        return c
    _render(build, hip, orc)


def test_analyser_behind_a_layout_above_stereo_follows_the_count_of_every_quantum(hip, orc):
    """analyser.rs:277-280 down-mixes the quantum it is handed: mono quanta as they are, stereo as 0.5 (L + R), quad as
    0.25 (L + R + SL + SR) — the analyser kernel reads the per-quantum codes in a dynamic-count plan"""
    taps = {}
    def build(be):
        c = _ctx(be)
        mono = _buf(c, 1, FRAMES, seed=41)
        quad = _buf(c, 4, RQ * 30, start=RQ * 70.25 / SR, seed=42)          # plays into the analyser's last window
        stereo = _buf(c, 2, RQ * 12, start=RQ * 80.0 / SR, seed=43)
        bus = c.create_gain(gain=0.8, channel_count=4, channel_count_mode="max", channel_interpretation="speakers")
        an = c.create_analyser(fft_size=2048, smoothing_time_constant=0.0)
        for s in (mono, quad, stereo):
            s.connect(bus)
        bus.connect(an).connect(c.create_biquad_filter(type_="highpass", frequency=500.0)).connect(c.destination())
        taps[id(be)] = an
        return c
    outs = {}
    for be in (hip, orc):
        c = build(be)
        if be is hip:
            assert "dynamic-count group" in c.plan_describe()
        outs[id(be)] = (c.start_rendering_sync().data, taps[id(be)].get_float_time_domain_data(instance=1),
                        taps[id(be)].get_float_frequency_data(instance=1))
        c.close()
    (g, gt, gf), (o, ot, of) = outs[id(hip)], outs[id(orc)]
    assert rms_err(g, o).max() <= 1e-6 and np.abs(gt - ot).max() <= 1e-6 and np.abs(ot).max() > 0.05
    fin = np.isfinite(of)
    assert np.abs(gf[fin] - of[fin]).max() <= 0.05


def _mixed_buffers(c, seed=70, rate=SR):
    """instance 0 plays a mono AudioBuffer, instance 1 a stereo one, instance 2 a 4-channel one (trimmed by the consumers)"""
    s = c.create_buffer_source()
    for i, nch in enumerate((1, 2, 4)):
        s.set_buffer(waa.AudioBuffer(white_noise(1, nch, FRAMES // 2 + 17 * i, seed0=seed + i)[0] * 0.5, rate), instance=i)
    s.start_at(130.0 / SR)
    return s


@pytest.mark.parametrize("consumer", ["stereo_panner", "biquad", "panner", "plain", "delay"])
def test_instances_play_buffers_of_different_channel_counts(hip, orc, consumer):
    """audio_buffer_source.rs:560-600: the output quantum has the channel count of the instance's OWN buffer; the static plan has
    one count per signal for the whole batch, so the planner answers with the dynamic-count plan whose codes are per instance
    (the mono instance takes the StereoPanner's / PannerNode's mono law and the mono->stereo up-mix at the destination)."""
    def build(be):
        c = _ctx(be)
        s = _mixed_buffers(c)
        if consumer == "stereo_panner":
            n = c.create_stereo_panner()
            n.pan.set_value(0.4)
        elif consumer == "biquad":
            n = c.create_biquad_filter()
        elif consumer == "panner":
            n = c.create_panner()
            n.position_x.set_value(1.5)
            n.position_z.set_value(-0.5)
        elif consumer == "delay":
            n = c.create_delay(0.1)
            n.delay_time.set_value(0.013)
        else:
            n = c.create_gain()
        s.connect(n)
        n.connect(c.destination())
        return c
    g, o = _render(build, hip, orc)
    assert np.abs(o[0, 0]).max() > 0.05 and np.abs(o[1, 1]).max() > 0.05
    if consumer == "plain":  # mono -> stereo copies (speakers), stereo stays
        assert np.array_equal(g[0, 0], g[0, 1]) and not np.array_equal(g[1, 0], g[1, 1])


def test_mixed_buffer_counts_with_a_resampled_buffer(hip, orc):
    """the same through the resampling source path (buffer rate != context rate)"""
    def build(be):
        c = _ctx(be)
        s = _mixed_buffers(c, seed=80, rate=44100.0)
        g = c.create_gain()
        g.gain.set_value(0.7)
        s.connect(g)
        g.connect(c.destination())
        return c
    _render(build, hip, orc)


@pytest.mark.measure
def test_mixed_buffer_counts_can_be_pinned_to_the_static_plan(hip):
    """WAA_STATIC_CHANNEL_COUNTS (measurement switch): the static plan has one channel count per signal -> status 4"""
    import os
    os.environ["WAA_STATIC_CHANNEL_COUNTS"] = "1"
    try:
        c = _ctx(hip)
        s = _mixed_buffers(c)
        s.connect(c.destination())
        with pytest.raises(waa.WaaError) as ei:
            c.start_rendering_sync()
        assert ei.value.status == 4 and "same channel count" in str(ei.value)
        c.close()
    finally:
        del os.environ["WAA_STATIC_CHANNEL_COUNTS"]


@pytest.mark.measure
def test_quantum_pipeline_is_bit_identical_to_the_one_wavefront_form(hip, monkeypatch):
    """dyn_kernel<2, W>: the items cut into W stages, stage w rendering quantum t - w in step t (round 5) — every item runs the
    code of the one-wavefront form on the same values, so the render must be BIT-identical (WAA_DYN_NO_PIPE=1 launches W = 1
    on the same plan).  Random graphs of both fuzz generators: those whose plan has a pipelined group are compared."""
    from test_fuzz_graphs import build_random_graph
    compared = 0
    staged = set()
    for seed in range(300, 420):
        for frozen in (False, True):
            ch, descr = build_random_graph(hip, seed, frozen=frozen)
            try:
                plan = ch.plan_describe()
            except waa.WaaError as e:
                if e.status == 4:
                    continue
                raise
            if "pipelined over the quanta" not in plan:
                ch.close()
                continue
            for line in plan.splitlines():
                if "pipelined over the quanta in" in line:
                    staged.add(int(line.split("pipelined over the quanta in ")[1].split()[0]))
            a = ch.start_rendering_sync().data
            ch.close()
            monkeypatch.setenv("WAA_DYN_NO_PIPE", "1")
            ch, _ = build_random_graph(hip, seed, frozen=frozen)
            b = ch.start_rendering_sync().data
            ch.close()
            monkeypatch.delenv("WAA_DYN_NO_PIPE")
            assert np.array_equal(a, b), (seed, frozen, descr, float(np.abs(a - b).max()))
            compared += 1
    print(f"{compared} random graphs with a pipelined dynamic-count group, stage counts seen: {sorted(staged)}")
    assert compared >= 20 and len(staged) >= 2
