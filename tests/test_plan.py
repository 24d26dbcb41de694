"""Host logic of the product without a GPU: plan-only batches (device = WAA_DEVICE_PLAN_ONLY) expose the
launch plan the planner / scheduler derive from a graph (chain fusion, streaming-biquad segments, convolver
block sizing, source schedules)."""
import numpy as np
import pytest

import web_audio_api_rs_amd as waa
from graphs import c2, c4, c5, garage_like_ir, t1, white_noise

RQ = 128


def plan(ctx):
    text = ctx.plan_describe()
    ctx.close()
    return text.strip().splitlines()


def test_c2_is_one_streaming_launch(hip):
    noise = white_noise(3, 2, 480000)
    ctx, _ = c2(hip, noise, device=waa.PLAN_ONLY)
    lines = plan(ctx)
    assert "3750 quanta, 235 tiles" in lines[0]
    assert any("quanta fast=3750 fast_loop=0 slow=0 silent=0 fast_tiles=234/235" in l for l in lines)
    assert any("1 distinct schedule(s) for 3 instance(s)" in l for l in lines)
    assert lines[-1] == "biquad_stream in=source:2ch gains=1 out=final"
    assert sum(l.startswith(("chain", "biquad_stream", "convolver")) for l in lines) == 1


def test_plan_only_batch_cannot_render(hip):
    ctx, _ = c2(hip, white_noise(1, 2, 1280), device=waa.PLAN_ONLY)
    with pytest.raises(waa.WaaError) as e:
        ctx.start_rendering_sync()
    assert e.value.status == 5 and "no CPU fallback" in str(e.value)


def test_t1_and_c4_plans(hip):
    noise, ir = white_noise(5, 2, 480000), garage_like_ir()
    ctx, _ = t1(hip, noise, ir, device=waa.PLAN_ONLY)
    lines = plan(ctx)
    # the Biquad in front of the long convolver has the same constant coefficients on every context: Biquad and Convolver are
    # LTI, the filter's transfer function is folded into the impulse response (round 5) — no launch, no signal, no per-sample work.
    # (garage_like_ir ends on a noise floor of 1e-3: the filter's ringing behind it needs a 23rd partition; the real response does not)
    assert not any(l.startswith("biquad_stream") for l in lines)
    assert any("folded into the impulse response (178899 -> " in l and "(its source is read in place)" in l for l in lines)
    conv = [l for l in lines if l.startswith("convolver")]
    assert len(conv) == 1 and "fft B=8192 N=16384 P=23 blocks=59 pairs=3 cin=2 cout=2 terms=2" in conv[0]
    assert "the Biquad in front, in the impulse response" in conv[0]
    assert lines[-1].startswith("alias node 0")  # the destination aliases the convolver output: no copy
    # per-context coefficients: per-context responses would be needed — the filter stays a stage of the forward transform
    ctx, nodes = t1(hip, noise, ir, device=waa.PLAN_ONLY, biquad_handle=True)
    nodes["biquad"].frequency.set_value(300.0, instance=2)
    lines = plan(ctx)
    assert any("filtered by the forward transform's input stage (its source is read in place)" in l for l in lines)
    conv = [l for l in lines if l.startswith("convolver")]
    assert "P=22 blocks=59" in conv[0] and "the Biquad in front, in the forward transform" in conv[0]
    ctx, _ = c4(hip, noise, ir, device=waa.PLAN_ONLY)
    lines = plan(ctx)
    assert "chain parallel C=2 in=[signal:2ch]->2ch ops=[STEREO_PAN] out=2ch" in lines
    assert lines[-1].startswith("alias node 0")  # analyser output == destination


@pytest.mark.measure
def test_t1_unfolded_plan_switch(hip, monkeypatch):
    """WAA_NO_CONV_BIQUAD_FOLD=1 keeps the Biquad a launch of its own (same-box A/B, and the parity tests' cross-check)."""
    monkeypatch.setenv("WAA_NO_CONV_BIQUAD_FOLD", "1")
    ctx, _ = t1(hip, white_noise(5, 2, 480000), garage_like_ir(), device=waa.PLAN_ONLY)
    lines = plan(ctx)
    assert "biquad_stream in=source:2ch gains=0 out=final" in lines
    assert not any("in the forward transform" in l or "in the impulse response" in l for l in lines)


@pytest.mark.measure
def test_t1_kernel_fold_switch(hip, monkeypatch):
    """WAA_NO_CONV_BIQUAD_IR_FOLD=1: the filter in front stays the exact-order stage of the forward transform (round 3's form;
    the A/B partner of the impulse-response fold and the form long-memory / per-context filters take)."""
    monkeypatch.setenv("WAA_NO_CONV_BIQUAD_IR_FOLD", "1")
    ctx, _ = t1(hip, white_noise(5, 2, 480000), garage_like_ir(), device=waa.PLAN_ONLY)
    lines = plan(ctx)
    assert any("filtered by the forward transform's input stage" in l for l in lines)
    assert any("P=22 blocks=59" in l and "in the forward transform" in l for l in lines)


@pytest.mark.parametrize("freq,q,folded", [(200.0, 1.0, True), (2000.0, 0.7, True), (30.0, 30.0, False), (20.0, 40.0, False)])
def test_impulse_response_fold_needs_a_short_filter_memory(hip, freq, q, folded):
    """a resonant low filter rings for tens of thousands of frames: its ringing does not die inside the two-block extension,
    the response fold is not taken (bandpass Q = 30 at 30 Hz: pole radius 0.99993)"""
    noise, ir = white_noise(2, 2, 480000), garage_like_ir()
    ctx = waa.OfflineAudioContext(2, 480000, 48000.0, n_instances=2, binding=hip, device=waa.PLAN_ONLY)
    src = ctx.create_buffer_source()
    src.set_buffer_batch(noise, 48000.0)
    bq = ctx.create_biquad_filter(type_="bandpass", frequency=freq, q=q)
    src.connect(bq).connect(ctx.create_convolver(buffer=waa.AudioBuffer(ir, 48000.0))).connect(ctx.destination())
    src.start()
    lines = plan(ctx)
    assert any("folded into the impulse response" in l for l in lines) == folded
    assert any("filtered by the forward transform's input stage" in l for l in lines) == (not folded)


def test_one_shared_audio_buffer_is_read_in_place(hip):
    """set_buffer for ALL instances = one AudioBuffer shared by every context (instance stride 0): read in place by a
    convolver like a per-instance batch (round-2 advisor finding: the view needed a strictly positive stride)."""
    ctx = waa.OfflineAudioContext(2, 480000, 48000.0, n_instances=4, binding=hip, device=waa.PLAN_ONLY)
    src = ctx.create_buffer_source()
    src.set_buffer(waa.AudioBuffer(white_noise(1, 2, 480000)[0], 48000.0))
    src.connect(ctx.create_convolver(buffer=waa.AudioBuffer(garage_like_ir(), 48000.0))).connect(ctx.destination())
    src.start()
    assert any("renders its AudioBuffer unchanged" in l and "reads it in place" in l for l in plan(ctx))


def test_c5_slow_track_goes_to_the_parallel_kernel(hip):
    ctx, _ = c5(hip, white_noise(2, 2, 5000), length=RQ * 50)
    ctx.device = waa.PLAN_ONLY
    lines = plan(ctx)
    assert any("fast=0 fast_loop=0 slow=50 silent=0 fast_tiles=0/4" in l for l in lines)
    assert lines[-1] == "chain parallel C=2 in=[source:2ch]->2ch ops=[WAVESHAPER] out=2ch"


@pytest.mark.parametrize("ir_len,expect", [(1, "direct FIR taps=1"), (128, "direct FIR taps=128"),
                                           (129, "fft B=128 N=256 P=2"), (3000, "fft B=128 N=256 P=24"),
                                           (3073, "fft B=512 N=1024 P=7"), (20000, "fft B=2048 N=4096 P=10"),
                                           (70000, "fft B=8192 N=16384 P=9"), (400000, "fft B=8192 N=16384 P=49")])
def test_convolver_block_sizing(hip, ir_len, expect):
    ir = np.ones((2, ir_len), np.float32)
    ctx, _ = t1(hip, white_noise(2, 2, RQ * 10), ir, with_biquad=False, device=waa.PLAN_ONLY)
    lines = plan(ctx)
    assert any(expect in l for l in lines), lines


def test_trailing_zeros_of_the_ir_are_trimmed(hip):
    """FFTConvolver::init drops trailing |h| < 1e-6 samples; an all-zero IR renders zeros."""
    ir = np.zeros((1, 5000), np.float32)
    ir[0, :40] = 1.0
    ctx, _ = t1(hip, white_noise(1, 1, RQ * 4), ir, with_biquad=False, device=waa.PLAN_ONLY)
    assert any("direct FIR taps=40" in l for l in plan(ctx))
    ctx, _ = t1(hip, white_noise(1, 1, RQ * 4), np.zeros((1, 64), np.float32), with_biquad=False, device=waa.PLAN_ONLY)
    assert any("all-zero impulse response" in l for l in plan(ctx))


def test_chains_are_split_around_constant_biquads(hip):
    sr = 48000.0
    ctx = waa.OfflineAudioContext(2, RQ * 40, sr, n_instances=2, binding=hip, device=waa.PLAN_ONLY)
    src = ctx.create_buffer_source()
    src.set_buffer_batch(white_noise(2, 1, RQ * 40), sr)
    g0, b1, g1 = ctx.create_gain(gain=0.9), ctx.create_biquad_filter(type_="peaking"), ctx.create_gain(gain=0.7)
    pan, b2 = ctx.create_stereo_panner(pan=-0.3), ctx.create_biquad_filter(type_="highpass")
    src.connect(g0).connect(b1).connect(g1).connect(pan).connect(b2).connect(ctx.destination())
    src.start()
    steps = [l for l in plan(ctx) if l.startswith(("chain", "biquad_stream"))]
    assert steps == [
        "chain parallel C=1 in=[source:1ch]->1ch ops=[GAIN] out=1ch",
        "biquad_stream in=signal:1ch gains=1 out=temp",
        "chain parallel C=2 in=[signal:1ch]->1ch ops=[STEREO_PAN] out=2ch",
        "biquad_stream in=signal:2ch gains=0 out=final",
    ]


def test_automated_biquads(hip):
    noise = white_noise(1, 2, RQ * 20)
    ctx, nodes = c2(hip, noise, device=waa.PLAN_ONLY)
    nodes["biquad"].frequency.set_block(0, np.linspace(100, 1000, 20).astype(np.float32))
    assert "biquad_stream(k-rate) in=source:2ch gains=1 out=final" in plan(ctx)  # per-quantum coefficients: streaming too
    ctx, nodes = c2(hip, noise, device=waa.PLAN_ONLY)
    nodes["biquad"].frequency.set_value_at_time(10.0, 0.0).exponential_ramp_to_value_at_time(10000.0, 0.05)
    # per-frame coefficients (a-rate automation, the same for every instance): one shared table, a lane per stream, tiles in
    # parallel (waa_biquad_lanes.hip) — what the device runs; a plan-only context describes the same launch list since round 4
    # (it used to fall back to the round-2 streaming form there: ADVICE round 3)
    p = plan(ctx)
    assert "biquad_lanes(a-rate, shared table: one lane per stream, tiles in parallel) in=source:2ch gains=1 out=final" in p
    assert "biquad_stream(a-rate" not in p


def test_fan_in_above_four_inputs_is_reduced_in_order(hip):
    sr = 44100.0
    ctx = waa.OfflineAudioContext(2, RQ * 8, sr, binding=hip, device=waa.PLAN_ONLY)
    for k in range(9):
        s = ctx.create_buffer_source()
        s.set_buffer(waa.AudioBuffer(np.ones((2, 64), np.float32), sr))
        s.connect(ctx.destination())
        s.start_at(k * 0.001)
    lines = plan(ctx)
    assert sum("fan-in partial sum of 4 inputs" in l for l in lines) == 2  # 9 -> 6 -> 3 inputs
    # the sources are fetched by the summing kernels themselves (no materialised copy of each source)
    assert not any(l.startswith("chain parallel C=2 in=[source:2ch]->") for l in lines)
    assert lines[-1].startswith("chain parallel C=2 in=[signal:2ch+source:2ch+source:2ch]")


def test_source_schedules_are_deduplicated_per_distinct_timing(hip):
    sr = 48000.0
    ctx = waa.OfflineAudioContext(1, RQ * 30, sr, n_instances=6, binding=hip, device=waa.PLAN_ONLY)
    src = ctx.create_buffer_source()
    src.set_buffer(waa.AudioBuffer(np.ones((1, RQ * 10), np.float32), sr))
    src.connect(ctx.destination())
    for i in range(6):
        src.start_at([0.0, 0.0, 1.5 / sr, 1.5 / sr, 0.01, 0.0][i], instance=i)
    lines = plan(ctx)
    assert any("3 distinct schedule(s) for 6 instance(s)" in l for l in lines)
    # aligned start: 10 fast quanta then silence; sub-sample start: slow track until the buffer ends
    assert any("quanta fast=10 fast_loop=0 slow=0 silent=20" in l for l in lines)
    assert any("fast=0 fast_loop=0 slow=11 silent=19" in l for l in lines)


def test_wide_filters_plan_on_the_streaming_kernel_and_the_remaining_limit_is_loud(hip):
    sr = 48000.0
    ctx = waa.OfflineAudioContext(4, RQ * 4, sr, binding=hip, device=waa.PLAN_ONLY)
    src = ctx.create_buffer_source()
    src.set_buffer(waa.AudioBuffer(np.ones((4, RQ * 4), np.float32), sr))
    src.connect(ctx.create_biquad_filter()).connect(ctx.destination())
    src.start()
    assert "biquad_stream in=source:4ch gains=0 out=final" in ctx.plan_describe()  # (refused with status 4 until round 3)
    ctx.close()
    # a channel-count CHANGE above stereo: rendered by the six-channel instantiation of the exact per-quantum path (round 3) ...
    def changing(with_analyser):
        ctx = waa.OfflineAudioContext(4, RQ * 8, sr, binding=hip, device=waa.PLAN_ONLY)
        mono, quad = ctx.create_buffer_source(), ctx.create_buffer_source()
        mono.set_buffer(waa.AudioBuffer(np.ones((1, RQ * 8), np.float32), sr))
        quad.set_buffer(waa.AudioBuffer(np.ones((4, RQ * 4), np.float32), sr))
        bq = ctx.create_biquad_filter()
        mono.connect(bq)
        quad.connect(bq)
        tail = bq.connect(ctx.create_delay(0.1, delay_time=0.01))
        if with_analyser:
            tail = tail.connect(ctx.create_analyser(fft_size=256))
        tail.connect(ctx.destination())
        mono.start()
        quad.start_at(RQ * 2 / sr)
        return ctx
    for with_analyser in (False, True):   # (DelayNodes re-mix their ring in place, the analyser kernel follows the per-quantum codes)
        ctx = changing(with_analyser)
        assert "dynamic-count group" in ctx.plan_describe()
        ctx.close()


# --------------------------------------------------------------------------- dynamic channel-count notes
def _plan_only(n_ch=2, frames=RQ * 40):
    return waa.OfflineAudioContext(n_ch, frames, 48000.0, n_instances=2, binding=waa.default_binding(), device=waa.PLAN_ONLY)


def _buffer(c, nch, frames, start=0.0):
    s = c.create_buffer_source()
    s.set_buffer_batch(np.zeros((2, nch, frames), np.float32), 48000.0)
    s.start_at(start)
    return s


def test_note_mono_first_then_stereo_into_a_filter():
    """the reference filters one channel until the stereo source starts (biquad_filter.rs:800-815)"""
    c = _plan_only()
    mono, stereo = _buffer(c, 1, RQ * 40), _buffer(c, 2, RQ * 40, start=RQ * 5 / 48000.0)
    bq = c.create_biquad_filter()
    mono.connect(bq)
    stereo.connect(bq)
    bq.connect(c.destination())
    plan = c.plan_describe()
    assert "narrower than its static channel count at quantum 0" in plan
    # the planner answers with the exact per-quantum channel counts of waa_dyn.hip: filter + destination in one group
    assert "dynamic-count group: 2 item(s) per quantum [BIQUAD3,pass0]" in plan and "dynamic channel count" not in plan
    c.close()


def test_buffers_of_different_channel_counts_across_instances_plan_dynamically():
    """audio_buffer_source.rs:560-600: each instance's source has its own buffer's channel count -> per-instance codes"""
    c = _plan_only()
    s = c.create_buffer_source()
    s.set_buffer(waa.AudioBuffer(np.zeros((1, RQ * 40), np.float32), 48000.0), instance=0)
    s.set_buffer(waa.AudioBuffer(np.zeros((2, RQ * 30), np.float32), 48000.0), instance=1)
    s.start_at(0.0)
    p = c.create_stereo_panner()
    s.connect(p)
    p.connect(c.destination())
    plan = c.plan_describe()
    assert "AudioBuffers of different channel counts -> per-instance counts through dyn_kernel" in plan, plan
    assert "dynamic-count group: 2 item(s) per quantum" in plan
    assert c.plan_describe() == plan  # (planning again finds the widened copies in place)
    c.close()


@pytest.mark.measure
def test_static_plan_switch_keeps_the_round1_note(monkeypatch):
    """WAA_STATIC_CHANNEL_COUNTS=1 (A/B aid): the static plan of round 1 with its 'dynamic channel count' note"""
    monkeypatch.setenv("WAA_STATIC_CHANNEL_COUNTS", "1")
    c = _plan_only()
    mono, stereo = _buffer(c, 1, RQ * 40), _buffer(c, 2, RQ * 40, start=RQ * 5 / 48000.0)
    bq = c.create_biquad_filter()
    mono.connect(bq)
    stereo.connect(bq)
    bq.connect(c.destination())
    plan = c.plan_describe()
    assert "dynamic channel count" in plan and "dynamic-count group" not in plan and "biquad_stream" in plan
    c.close()


def test_no_note_when_widths_agree_over_time():
    """same graph, both sources from t = 0: the count never changes; a lone stereo source that ends early keeps the
    filter's count while it rings (biquad_filter.rs:786-798)"""
    c = _plan_only()
    mono, stereo = _buffer(c, 1, RQ * 40), _buffer(c, 2, RQ * 40)
    bq = c.create_biquad_filter()
    mono.connect(bq)
    stereo.connect(bq)
    bq.connect(c.destination())
    assert "dynamic-count group" not in c.plan_describe()
    c.close()
    c = _plan_only()
    short = _buffer(c, 2, RQ * 3)
    bq = c.create_biquad_filter()
    short.connect(bq).connect(c.destination())
    assert "dynamic-count group" not in c.plan_describe()
    c.close()


def test_note_delay_line_collapses_when_its_stereo_input_ends():
    """delay.rs:469-489: the line is re-mixed to the channel count of the current (silent = mono) input"""
    c = _plan_only()
    short = _buffer(c, 2, RQ * 3)
    d = c.create_delay(0.1, delay_time=0.05)
    short.connect(d).connect(c.destination())
    plan = c.plan_describe()
    assert "falls silent (= mono) while the node still holds multi-channel material" in plan
    assert "dynamic-count group: 3 item(s) per quantum [delayW2,delayR2,pass0]" in plan
    c.close()
    # a mono input into the same delay never changes the count
    c = _plan_only()
    short = _buffer(c, 1, RQ * 3)
    d = c.create_delay(0.1, delay_time=0.05)
    short.connect(d).connect(c.destination())
    assert "dynamic-count group" not in c.plan_describe()
    c.close()


@pytest.mark.measure
def test_note_zero_gain_in_front_of_a_panner_and_strict_mode(monkeypatch):
    """gain.rs:163-170: |g| <= 1e-6 emits a silent (mono) quantum; the StereoPanner behind it then sees mono"""
    def build():
        c = _plan_only()
        a, b2 = _buffer(c, 2, RQ * 40), _buffer(c, 1, RQ * 40)
        g = c.create_gain(gain=1.0)
        vals = np.ones(40, np.float32)
        vals[10:20] = 0.0
        g.gain.set_block(0, vals)
        pan = c.create_stereo_panner(pan=0.3)
        a.connect(g).connect(pan)
        b2.connect(pan)
        pan.connect(c.destination())
        return c
    c = build()
    plan = c.plan_describe()
    assert "narrower than its static channel count at quantum 10" in plan and "dynamic-count group" in plan
    c.close()
    monkeypatch.setenv("WAA_STATIC_CHANNEL_COUNTS", "1")
    monkeypatch.setenv("WAA_STRICT_CHANNEL_COUNTS", "1")
    c = build()
    with pytest.raises(waa.WaaError) as ei:
        c.plan_describe()
    assert ei.value.status == 4
    c.close()


@pytest.mark.measure
def test_plan_validator_refuses_a_misordered_launch_list(hip, monkeypatch):
    """build_plan ends with a read-before-write check over the launch list (a consumer launched before its producer
    would render stale data silently).  WAA_DEBUG_REVERSE_PLAN reverses the list before the check: the dependent
    launches of a Biquad -> Convolver -> StereoPanner graph must not get past it."""
    def build():
        ctx = waa.OfflineAudioContext(2, RQ * 40, 48000.0, n_instances=2, binding=hip, device=waa.PLAN_ONLY)
        src = ctx.create_buffer_source()
        src.set_buffer_batch(white_noise(2, 2, RQ * 40), 48000.0)
        bq = ctx.create_biquad_filter(type_="lowpass", frequency=500.0)
        conv = ctx.create_convolver(buffer=waa.AudioBuffer(white_noise(1, 2, 300)[0], 48000.0))
        pan = ctx.create_stereo_panner(pan=0.3)
        src.connect(bq).connect(conv).connect(pan).connect(ctx.destination())
        src.start()
        return ctx
    assert len(plan(build())) >= 3
    monkeypatch.setenv("WAA_DEBUG_REVERSE_PLAN", "1")
    with pytest.raises(waa.WaaError, match="reads a buffer that a later launch produces") as ei:
        build().plan_describe()
    assert ei.value.status == 3


# --------------------------------------------------------------------------- round-2 folds (plan text only, no GPU)
def _ctx(hip, n_ch=2, frames=2048 * 8, n=3):
    c = waa.OfflineAudioContext(2, frames, 48000.0, n_instances=n, binding=hip, device=waa.PLAN_ONLY)
    s = c.create_buffer_source()
    s.set_buffer_batch(white_noise(n, n_ch, frames), 48000.0)
    s.start()
    return c, s


def test_echo_is_one_launch(hip):
    """source -> [dry] + [Delay -> Gain] -> destination: the buffer is read in place by both consumers, the delay from its
    line, the gain on the edge — one chain launch"""
    c, s = _ctx(hip)
    s.connect(c.destination())
    s.connect(c.create_delay(1.0, delay_time=0.05)).connect(c.create_gain(gain=0.5)).connect(c.destination())
    lines = plan(c)
    assert any("its 2 consumers read it in place" in l for l in lines)
    assert any("read by its consumers from the delay line" in l for l in lines)
    assert lines[-1] == "chain parallel C=2 in=[signal:2ch+gain*delayed:2ch]->2ch ops=[] out=2ch"
    assert sum(l.startswith(("chain", "biquad_stream", "delay node 2: 2ch delayTime")) for l in lines) == 1


def test_echo_keeps_the_gather_kernel_for_a_convolver_consumer(hip):
    c, s = _ctx(hip)
    d = c.create_delay(1.0, delay_time=0.05)
    s.connect(d).connect(c.create_convolver(buffer=waa.AudioBuffer(np.ones((1, 300), np.float32), 48000.0))).connect(c.destination())
    lines = plan(c)
    assert any(l.startswith("delay node") and "ring=" in l for l in lines)
    assert not any("delayed:" in l for l in lines)


def test_lfo_depth_gain_rides_on_the_param_edge(hip):
    """Oscillator -> Gain(depth) -> carrier.frequency: the depth is an edge gain of the param's summing chain"""
    c = waa.OfflineAudioContext(2, 2048 * 4, 48000.0, n_instances=2, binding=hip, device=waa.PLAN_ONLY)
    lfo = c.create_oscillator(frequency=5.0)
    depth = c.create_gain(gain=30.0)
    car = c.create_oscillator(frequency=440.0)
    lfo.connect(depth).connect(car.frequency)
    car.connect(c.destination())
    lfo.start()
    car.start()
    lines = plan(c)
    assert any("gain node" in l and "folded into an input edge" in l for l in lines)
    assert any("in=[gain*signal:1ch]->1ch ops=[PARAM_ADD]" in l for l in lines)
    assert sum("ops=[GAIN]" in l for l in lines) == 0


def test_short_delay_loop_uses_the_quantum_serial_kernel(hip):
    """a loop delay below one 2048-frame tile cannot be block-scheduled: the quantum-serial kernel renders it — unless the loop has
    the ring kernel's shape (Delay <-> Gain [-> constant Biquad], delay >= 136 frames), which walks it in chunks down to 128 frames (round 4)"""
    c, s = _ctx(hip)
    d = c.create_delay(1.0, delay_time=0.00275)   # 132 frames: below the ring kernel's smallest chunk (128 + 8)
    s.connect(d)
    d.connect(c.create_gain(gain=0.5)).connect(d)
    d.connect(c.destination())
    lines = plan(c)
    assert any(l.startswith("feedback loop:") and "item(s) per quantum" in l for l in lines)
    assert not any("block-scheduled" in l for l in lines)
    c, s = _ctx(hip)
    d = c.create_delay(1.0, delay_time=0.01)    # 480 frames: the ring kernel, a 1024-frame ring
    s.connect(d)
    d.connect(c.create_gain(gain=0.5)).connect(d)
    d.connect(c.destination())
    lines = plan(c)
    assert any("LDS-ring kernel in ONE launch" in l and "chunks of 256 frames" in l and "last 1024 frames" in l for l in lines), lines
    assert not any("item(s) per quantum" in l for l in lines)
    c, s = _ctx(hip)
    d = c.create_delay(1.0, delay_time=0.01)    # the same delay around a WaveShaper: not the ring kernel's shape
    s.connect(d)
    d.connect(c.create_wave_shaper(curve=np.float32([-1.0, 0.0, 1.0]))).connect(c.create_gain(gain=0.5)).connect(d)
    d.connect(c.destination())
    lines = plan(c)
    assert any(l.startswith("feedback loop:") and "item(s) per quantum" in l for l in lines)
    assert not any("LDS-ring" in l for l in lines)


def test_pipeline_stages_never_cut_a_delay_pair_or_a_loop(hip):
    """plan-level: a DelayNode's writer and reader (and a feedback loop's members) share a stage"""
    FRAMES, SR, N = 128 * 90 + 50, 48000.0, 3
    c = waa.OfflineAudioContext(2, FRAMES, SR, n_instances=N, binding=hip, device=waa.PLAN_ONLY)
    def _buf(c, nch, frames, start=0.0, seed=1):
        s = c.create_buffer_source()
        s.set_buffer_batch(white_noise(N, nch, frames, seed0=seed) * 0.5, SR)
        s.start_at(start)
        return s
    mono = _buf(c, 1, FRAMES, seed=5)
    st = _buf(c, 2, FRAMES // 2, start=0.3 * FRAMES / SR, seed=6)
    g0 = c.create_gain(gain=0.9)
    d = c.create_delay(0.1, delay_time=0.01)
    fb = c.create_gain(gain=0.4)
    pan = c.create_stereo_panner(pan=-0.2)
    bq = c.create_biquad_filter(type_="lowpass", frequency=900.0)
    mono.connect(g0)
    st.connect(g0)
    g0.connect(d)
    d.connect(fb).connect(d)
    d.connect(pan).connect(bq).connect(c.destination())
    plan = c.plan_describe()
    line = next(l for l in plan.splitlines() if l.startswith("dynamic-count group"))
    assert "pipelined over the quanta" in line, plan
    items = line.split("[")[1].split("]")[0].split(",")
    # cuts are named by the third of an item they stand in front of: "2" = item 2's gather + mix, "2b" = its node, "2c" = its publication
    cuts = [3 * int(x.rstrip("bc")) + (1 if x.endswith("b") else 2 if x.endswith("c") else 0) for x in line.rsplit("item(s) ", 1)[1].split(",")]
    # the loop (delayR .. delayW and the gain between) is one contiguous run without a cut inside
    loop_idx = [i for i, t in enumerate(items) if t.startswith("delay") or t.startswith("GAIN" + str(fb.id))]
    lo, hi = min(loop_idx), max(loop_idx)
    assert not any(3 * lo < u <= 3 * hi + 2 for u in cuts), (items, cuts)
    assert len(cuts) >= 1
