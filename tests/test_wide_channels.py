"""Signals of 7 ... 32 channels (MAX_CHANNELS, src/lib.rs:21): `AudioRenderQuantum::mix` treats every mix with more than six channels
on either side as "discrete" — pad with silence / truncate — whatever the node's interpretation says (src/render/quantum.rs:285-306).

The reference's own integration tests (tests/mixing.rs) stop at quad; these are the same cases — a ConstantSource through a GainNode
with a channel configuration into a destination with one — at 7, 8, 16 and 32 channels, plus what a multi-channel render actually looks
like: AudioBuffers of N channels through gains, filters, delays, curves, sums of signals of different widths, and the nodes that clamp
their input to stereo.  CPU: the oracle against hand-derived expectations; GPU: the library against the oracle (round 6: chains on wide
signals are rendered in channel slices of six, waa_plan.cpp::push_chain_step; the streaming filter kernels run one wavefront per channel
and always took any count)."""
import numpy as np
import pytest

import web_audio_api_rs_amd as waa

RQ = 128
SR = 44100.0
WIDTHS = [7, 8, 16, 32]


def rms(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


def _run_mixing(be, n_out, dest_interp, count, mode, interp, n_inst=1):
    """tests/mixing.rs:9-37"""
    c = waa.OfflineAudioContext(n_out, 128, SR, n_instances=n_inst, binding=be)
    c.destination().set_channel_interpretation(dest_interp)
    k = c.create_constant_source()
    k.start()
    g = c.create_gain()
    g.set_channel_count(count)
    g.set_channel_count_mode(mode)
    g.set_channel_interpretation(interp)
    k.connect(g).connect(c.destination())
    out = c.start_rendering_sync().data
    c.close()
    return out


MIX_CASES = [  # (destination channels, destination interpretation, gain count, gain mode, gain interpretation) -> channels that carry the 1.0
    (8, "speakers", 8, "explicit", "speakers", [0]),       # mono -> 8: above six channels "speakers" is discrete: channel 0 only
    (8, "discrete", 8, "explicit", "discrete", [0]),
    (8, "speakers", 1, "max", "speakers", [0]),            # mono into an 8-channel destination: padded, not spread
    (8, "speakers", 2, "explicit", "speakers", [0, 1]),    # mono -> stereo by the speakers rule (copy), stereo -> 8 padded
    (8, "speakers", 6, "explicit", "speakers", [2]),       # mono -> 5.1 goes to the centre channel, 5.1 -> 8 padded
    (32, "speakers", 32, "explicit", "speakers", [0]),
    (32, "speakers", 4, "explicit", "speakers", [0, 1]),   # mono -> quad: L and R
    (16, "discrete", 32, "explicit", "speakers", [0]),     # 32 -> 16: truncated
    (2, "speakers", 8, "explicit", "speakers", [0]),       # 8 -> 2: truncated (NOT the 5.1 down-mix), channel 1 stays silent
    (1, "speakers", 7, "explicit", "discrete", [0]),
    (7, "speakers", 7, "clamped-max", "speakers", [0]),    # clamped-max 7 of a mono input = 1 channel, padded to 7 at the destination
]


@pytest.mark.parametrize("case", MIX_CASES, ids=lambda c: f"{c[0]}{c[1][0]}-{c[2]}{c[3][0]}{c[4][0]}")
def test_mixing_integration_wide(be, case):
    n_out, dest_interp, count, mode, interp, ones = case
    o = _run_mixing(be, n_out, dest_interp, count, mode, interp, n_inst=2)
    assert o.shape == (2, n_out, 128)
    for c in range(n_out):
        assert np.array_equal(o[0, c], np.full(128, 1.0 if c in ones else 0.0, np.float32)), (c, o[0, c, :4])
        assert np.array_equal(o[1, c], o[0, c])


def _noise(n_inst, n_ch, frames, seed):
    return np.random.default_rng(seed).uniform(-1, 1, (n_inst, n_ch, frames)).astype(np.float32)


def g_gain(ctx, src, n):
    g = ctx.create_gain(gain=0.5)
    for i in range(ctx.n_instances):
        g.gain.set_value(0.25 + 0.2 * i, instance=i)
    src.connect(g).connect(ctx.destination())


def g_gain_ramp(ctx, src, n):
    g = ctx.create_gain(gain=0.0)
    g.gain.linear_ramp_to_value_at_time(1.0, 0.02)
    src.connect(g).connect(ctx.destination())


def g_biquad(ctx, src, n):
    bq = ctx.create_biquad_filter(type_="lowpass", frequency=900.0, q=2.0)
    for i in range(ctx.n_instances):
        bq.frequency.set_value(500.0 + 700.0 * i, instance=i)
    src.connect(bq).connect(ctx.destination())


def g_biquad_gain(ctx, src, n):
    bq = ctx.create_biquad_filter(type_="peaking", frequency=1500.0, q=1.0, gain=6.0)
    g = ctx.create_gain(gain=0.7)
    src.connect(bq).connect(g).connect(ctx.destination())
    src.connect(ctx.destination())


def g_biquad_arate(ctx, src, n):
    bq = ctx.create_biquad_filter(type_="lowpass", frequency=400.0)
    bq.frequency.linear_ramp_to_value_at_time(5000.0, 0.03)
    src.connect(bq).connect(ctx.destination())


def g_iir(ctx, src, n):
    f = ctx.create_iir_filter([0.2, 0.3, 0.1], [1.0, -0.5, 0.2])
    src.connect(f).connect(ctx.destination())


def g_shaper(ctx, src, n):
    sh = ctx.create_wave_shaper()
    sh.set_curve(np.tanh(np.linspace(-2, 2, 257)).astype(np.float32) + 0.05)  # (curve(0) != 0: silence is shaped too)
    src.connect(sh).connect(ctx.destination())


def g_shaper_2x(ctx, src, n):
    sh = ctx.create_wave_shaper()
    sh.set_curve(np.tanh(np.linspace(-2, 2, 257)).astype(np.float32))
    sh.set_oversample("2x")
    src.connect(sh).connect(ctx.destination())


def g_shaper_4x(ctx, src, n):
    sh = ctx.create_wave_shaper()
    sh.set_curve((np.tanh(np.linspace(-2, 2, 65)) + 0.1).astype(np.float32))
    sh.set_oversample("4x")
    src.connect(sh).connect(ctx.destination())


def g_hrtf(ctx, src, n):
    p = ctx.create_panner(panning_model="HRTF", position=(1.0, 0.5, -0.3))
    src.connect(p).connect(ctx.destination())


def g_delay(ctx, src, n):
    d = ctx.create_delay(0.1)
    d.delay_time.set_value(0.0123)
    src.connect(d).connect(ctx.destination())
    src.connect(ctx.destination())


def g_mixed_widths(ctx, src, n):
    """a wide, a stereo and a mono signal summed at a GainNode (count mode max: the wide one decides; the narrow ones are padded)"""
    st = ctx.create_buffer_source()
    st.set_buffer_batch(_noise(ctx.n_instances, 2, ctx.length, 7), SR)
    st.start()
    k = ctx.create_constant_source(offset=0.3)
    k.start()
    g = ctx.create_gain(gain=0.8)
    src.connect(g)
    st.connect(g)
    k.connect(g)
    g.connect(ctx.destination())


def g_explicit_wider(ctx, src, n):
    """the wide signal into a node with a WIDER explicit count, then into a narrower destination input"""
    g = ctx.create_gain(gain=0.9, channel_count=min(32, n + 3), channel_count_mode="explicit", channel_interpretation="speakers")
    src.connect(g).connect(ctx.destination())


def g_stereo_nodes(ctx, src, n):
    """nodes that clamp their input to two channels hear channels 0 and 1 of a wide signal (discrete truncation, not a down-mix)"""
    p = ctx.create_stereo_panner(pan=0.4)
    cv = ctx.create_convolver(buffer=waa.AudioBuffer(_noise(1, 2, 300, 9)[0] * 0.1, SR), disable_normalization=True)
    src.connect(p).connect(ctx.destination())
    src.connect(cv).connect(ctx.destination())


def g_five_inputs(ctx, src, n):
    """more inputs than one launch sums (MAX_INPUTS = 4): the partial sum is a wide signal too"""
    for j in range(5):
        g = ctx.create_gain(gain=0.1 * (j + 1))
        src.connect(g).connect(ctx.destination())


GRAPHS = [g_gain, g_gain_ramp, g_biquad, g_biquad_gain, g_biquad_arate, g_iir, g_shaper, g_shaper_2x, g_shaper_4x, g_hrtf, g_delay, g_mixed_widths,
          g_explicit_wider, g_stereo_nodes, g_five_inputs]


def _render(be, graph, n, n_out, n_inst=3, length=20 * RQ + 37, start=0.0):
    ctx = waa.OfflineAudioContext(n_out, length, SR, n_instances=n_inst, binding=be)
    src = ctx.create_buffer_source()
    src.set_buffer_batch(_noise(n_inst, n, length + 64, 100 + n), SR)
    src.start_at(start)
    graph(ctx, src, n)
    out = ctx.start_rendering_sync().data
    ctx.close()
    return out


@pytest.mark.parametrize("n", [8, 32])
@pytest.mark.parametrize("graph", [g_gain, g_biquad, g_mixed_widths, g_stereo_nodes], ids=lambda g: g.__name__)
def test_wide_graphs_run_on_the_oracle(orc, graph, n):
    """(CPU) the oracle renders them, every channel of the wide source reaches its own output channel and nothing else does"""
    o = _render(orc, graph, n, n)
    assert o.shape[1] == n and np.isfinite(o).all()
    if graph is g_gain:
        x = _noise(3, n, 20 * RQ + 37 + 64, 100 + n)[:, :, :20 * RQ + 37]
        for i in range(3):
            assert np.array_equal(o[i], x[i] * np.float32(0.25 + 0.2 * i))
    if graph is g_stereo_nodes:
        assert np.abs(o[:, :2]).max() > 0.1 and np.all(o[:, 2:] == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("n", WIDTHS)
@pytest.mark.parametrize("graph", GRAPHS, ids=lambda g: g.__name__)
def test_wide_graphs_against_the_oracle(hip, orc, graph, n):
    try:
        g = _render(hip, graph, n, n)
    except waa.WaaError as e:
        # (until the end of round 6 oversampled WaveShapers on wide signals were refused here: nothing is any more)
        raise AssertionError(f"refused: {e}")
    o = _render(orc, graph, n, n)
    assert g.shape == o.shape and np.isfinite(g).all()
    assert np.abs(o).max() > 0.05
    for i in range(g.shape[0]):
        for c in range(n):
            assert rms(g[i, c], o[i, c]) <= 1e-6, (i, c, rms(g[i, c], o[i, c]))


@pytest.mark.gpu
@pytest.mark.parametrize("n,n_out", [(8, 2), (8, 6), (16, 8), (32, 7), (6, 8), (2, 32), (1, 16)])
def test_destination_narrower_or_wider_than_the_signal(hip, orc, n, n_out):
    """truncation / padding at the destination (its count is explicit): 8 -> 6 and 8 -> 2 are truncations, not 5.1 down-mixes"""
    g = _render(hip, g_gain, n, n_out)
    o = _render(orc, g_gain, n, n_out)
    assert g.shape == o.shape == (3, n_out, 20 * RQ + 37)
    assert np.array_equal(g, o)


@pytest.mark.gpu
def test_a_wide_source_that_starts_late_in_front_of_a_count_sensitive_node(hip, orc):
    """a wide source that starts late or ends early changes the reference's channel count mid-render (silence is mono); where nothing
    downstream is count-sensitive the static plan renders it, where something is (a DelayNode: its line is re-mixed — and loses
    channels 1 ... 7 of what it holds when the input falls silent) the exact per-quantum counts of the dynamic plan do (dyn_kernel<32>)"""
    g = _render(hip, g_gain, 8, 8, start=0.01)
    o = _render(orc, g_gain, 8, 8, start=0.01)
    assert np.array_equal(g, o)
    try:
        g = _render(hip, g_delay, 8, 8, start=0.01)
    except waa.WaaError as e:
        assert e.status == 4 and "wider than six channels" in str(e)
    else:
        o = _render(orc, g_delay, 8, 8, start=0.01)
        assert max(rms(g[i, c], o[i, c]) for i in range(3) for c in range(8)) <= 1e-6


@pytest.mark.gpu
def test_32_channels_at_batch_size(hip, orc):
    """256 contexts x 32 channels x 2 s through a filter and a gain: 8192 streams; every 16th context against the oracle"""
    n_inst, n, length = 256, 32, 750 * RQ
    x = _noise(n_inst, n, length, 5)

    def render(be, insts):
        ctx = waa.OfflineAudioContext(n, length, 48000.0, n_instances=len(insts), binding=be)
        src = ctx.create_buffer_source()
        src.set_buffer_batch(x[insts], 48000.0)
        src.start()
        bq = ctx.create_biquad_filter(type_="highpass", frequency=300.0)
        g = ctx.create_gain(gain=0.5)
        src.connect(bq).connect(g).connect(ctx.destination())
        out = ctx.start_rendering_sync().data
        ctx.close()
        return out

    g = render(hip, list(range(n_inst)))
    sample = list(range(0, n_inst, 16))
    o = render(orc, sample)
    for k, i in enumerate(sample):
        for c in range(n):
            assert rms(g[i, c], o[k, c]) <= 1e-6


# ---- the ORDER of a node's inputs (graph.rs:524-535, quantum.rs:425-470): the input bus grows input by input, and an earlier input is
# carried through every width the bus takes behind it — mono -> stereo -> 5.1 leaves a mono signal in L and R, mono -> 5.1 puts it in C
def _ordered_sum(be, order, wide, n_out, stereo_frames=None, n_inst=2, length=12 * RQ):
    """`order` = the order in which the reference sums the three inputs.  That is the order of its edge list, which both back-ends
    keep sorted by producer id, highest first: the producers are created in reverse"""
    ctx = waa.OfflineAudioContext(n_out, length, SR, n_instances=n_inst, binding=be)
    nodes = {}
    for name in reversed(order):
        if name == "mono":
            nodes[name] = ctx.create_constant_source(offset=0.3)
        else:
            nodes[name] = ctx.create_buffer_source()
            ch, seed, frames = (2, 21, stereo_frames or length) if name == "stereo" else (wide, 22, length)
            nodes[name].set_buffer_batch(_noise(n_inst, ch, frames, seed) * 0.1, SR)
        nodes[name].start()
    g = ctx.create_gain(gain=1.0)
    for name in order:
        nodes[name].connect(g)
    g.connect(ctx.destination())
    plan = ctx.plan_describe() if be.prefix != "orc_" else ""
    out = ctx.start_rendering_sync().data
    ctx.close()
    return out, plan


@pytest.mark.parametrize("wide", [4, 6, 8, 16])
def test_the_order_of_the_inputs_matters_on_the_oracle(orc, wide):
    """(CPU) with the mono source added first and a stereo one behind it the constant sits in channels 0 AND 1 of the sum, whatever comes
    later — also for a 5.1 bus, where a direct mono -> 5.1 mix would put it in the centre channel alone"""
    o, _ = _ordered_sum(orc, ["mono", "stereo", "wide"], wide, wide)
    x_st = _noise(2, 2, 12 * RQ, 21) * 0.1
    x_w = _noise(2, wide, 12 * RQ, 22) * 0.1
    const_in = (o[0] - x_w[0])
    const_in[:2] -= x_st[0]
    where = [c for c in range(wide) if np.abs(const_in[c]).max() > 0.2]
    assert where == [0, 1], where
    # ... and with the wide signal first the bus has its final width when the mono source arrives: its direct mix
    o, _ = _ordered_sum(orc, ["wide", "stereo", "mono"], wide, wide)
    const_in = (o[0] - x_w[0])
    const_in[:2] -= x_st[0]
    where = [c for c in range(wide) if np.abs(const_in[c]).max() > 0.2]
    assert where == ([0, 1] if wide == 4 else [2] if wide == 6 else [0]), where


@pytest.mark.gpu
@pytest.mark.parametrize("wide", [4, 6, 8, 32])
def test_the_order_of_the_inputs_static(hip, orc, wide):
    g, plan = _ordered_sum(hip, ["mono", "stereo", "wide"], wide, wide)
    o, _ = _ordered_sum(orc, ["mono", "stereo", "wide"], wide, wide)
    if wide > 4:  # (mono -> stereo -> quad IS mono -> quad: nothing to pre-mix)
        assert "is mixed along the widths the reference's input bus takes behind it" in plan, plan
    for i in range(g.shape[0]):
        for c in range(wide):
            assert rms(g[i, c], o[i, c]) <= 1e-6, (i, c)


@pytest.mark.gpu
def test_the_order_of_the_inputs_when_the_stereo_source_ends_early(hip, orc):
    """... and while the stereo source is silent the bus goes mono -> 5.1 directly: the constant moves to the centre channel.  That is a
    channel-count change mid-render: the exact per-quantum counts of the dynamic plan"""
    g, plan = _ordered_sum(hip, ["mono", "stereo", "wide"], 6, 6, stereo_frames=5 * RQ)
    o, _ = _ordered_sum(orc, ["mono", "stereo", "wide"], 6, 6, stereo_frames=5 * RQ)
    assert "dyn_kernel" in plan
    x_w = _noise(2, 6, 12 * RQ, 22) * 0.1
    assert np.abs((o[0] - x_w[0])[2, 6 * RQ:]).max() > 0.2 and np.abs((o[0] - x_w[0])[0, 6 * RQ:]).max() < 1e-6  # centre, not L
    for i in range(g.shape[0]):
        for c in range(6):
            assert rms(g[i, c], o[i, c]) <= 1e-6, (i, c)
    # ... and the same above six channels (dyn_kernel<32>, round 6): while the stereo source plays the constant is on channels 0 and 1,
    # afterwards on channel 0 alone (mono -> 8 is a padding)
    for wide in (8, 32):
        g, plan = _ordered_sum(hip, ["mono", "stereo", "wide"], wide, wide, stereo_frames=5 * RQ)
        o, _ = _ordered_sum(orc, ["mono", "stereo", "wide"], wide, wide, stereo_frames=5 * RQ)
        assert "dyn_kernel" in plan
        x_w = _noise(2, wide, 12 * RQ, 22) * 0.1
        assert np.abs((o[0] - x_w[0])[0, 6 * RQ:]).max() > 0.2 and np.abs((o[0] - x_w[0])[1, 6 * RQ:]).max() < 1e-6
        for i in range(g.shape[0]):
            for c in range(wide):
                assert rms(g[i, c], o[i, c]) <= 1e-6, (wide, i, c)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [8, 32])
def test_analyser_behind_a_wide_signal(hip, orc, n):
    """the AnalyserNode passes its (wide) input through and analyses its down-mix to mono — above six channels: channel 0
    (analyser.rs:265-290; quantum.rs:285-306)"""
    outs = []
    for be in (hip, orc):
        ctx = waa.OfflineAudioContext(n, 24 * RQ, SR, n_instances=2, binding=be)
        src = ctx.create_buffer_source()
        src.set_buffer_batch(_noise(2, n, 24 * RQ, 31), SR)
        src.start()
        an = ctx.create_analyser(fft_size=256)
        src.connect(an).connect(ctx.destination())
        data = ctx.start_rendering_sync().data
        outs.append((data, an.get_float_time_domain_data_all(), an.get_float_frequency_data_all()))
        ctx.close()
    (g, gt, gf), (o, ot, of) = outs
    assert np.array_equal(g, o)
    x = _noise(2, n, 24 * RQ, 31)
    assert np.array_equal(ot[0], x[0, 0, -256:])  # channel 0, not an average
    assert np.array_equal(gt, ot)
    loud = of > of.max() - 60.0
    assert np.abs(gf[loud] - of[loud]).max() <= 0.035


@pytest.mark.gpu
def test_render_sharded_with_eight_channel_buffers(hip, orc):
    """the host-to-host path (waa_render_sharded) with 7.1 AudioBuffers"""
    from web_audio_api_rs_amd.sharding import render_sharded
    n_inst, n, length = 12, 8, 40 * RQ
    x = _noise(n_inst, n, length, 41)

    def build(be):
        def f(count, device):
            ctx = waa.OfflineAudioContext(n, length, 48000.0, n_instances=count, binding=be, device=device)
            src = ctx.create_buffer_source()
            bq = ctx.create_biquad_filter(type_="lowpass", frequency=2000.0)
            src.connect(bq).connect(ctx.destination())
            src.start()
            return ctx, src
        return f

    out = np.zeros((n_inst, n, length), np.float32)
    render_sharded(build(hip), x, out, devices=[0], sub_batches=3, sample_rate=48000.0)
    ctx, src = build(orc)(n_inst, 0)
    src.set_buffer_batch(x, 48000.0)
    o = ctx.start_rendering_sync().data
    ctx.close()
    for i in range(n_inst):
        for c in range(n):
            assert rms(out[i, c], o[i, c]) <= 1e-6


def _echo_loop(be, n, delay_s, with_filter, burst, n_inst=2, length=60 * RQ):
    ctx = waa.OfflineAudioContext(n, length, SR, n_instances=n_inst, binding=be)
    src = ctx.create_buffer_source()
    src.set_buffer_batch(_noise(n_inst, n, 10 * RQ if burst else length, 51), SR)  # a burst, then its echoes (= silence = mono) / a signal
    src.start()
    d = ctx.create_delay(1.0)
    d.delay_time.set_value(delay_s)
    fb = ctx.create_gain(gain=0.6)
    src.connect(d)
    if with_filter:
        bq = ctx.create_biquad_filter(type_="lowpass", frequency=3000.0)
        d.connect(bq).connect(fb).connect(d)
    else:
        d.connect(fb).connect(d)
    d.connect(ctx.destination())
    src.connect(ctx.destination())
    plan = ctx.plan_describe() if be.prefix != "orc_" else ""
    out = ctx.start_rendering_sync().data
    ctx.close()
    return out, plan


@pytest.mark.gpu
@pytest.mark.parametrize("with_filter", [False, True], ids=["gain", "biquad"])
@pytest.mark.parametrize("delay_s", [0.004, 0.11], ids=["short", "long"])
@pytest.mark.parametrize("burst", [True, False], ids=["burst", "steady"])
@pytest.mark.parametrize("n", [4, 6, 8, 16])
def test_feedback_loops_on_wide_signals(hip, orc, n, burst, delay_s, with_filter):
    """an echo loop on quad / 5.1 / wider signals.  The quantum-serial loop kernel is mono / stereo; wider loops go to the dynamic-count
    kernel (<= 5.1) or, when every delay across the cut is long enough, to block-scheduled node-major launches (any width).  What is
    left refuses with status 4: above 5.1, a source that ends (silence is mono: the line is re-mixed) and delays shorter than a block"""
    try:
        g, plan = _echo_loop(hip, n, delay_s, with_filter, burst)
    except waa.WaaError as e:
        assert e.status == 4, e
        assert n > 6 and (burst or delay_s < 0.05), f"{n} channels, delay {delay_s}: {e}"
        pytest.skip(f"refused: {e}")
    o, _ = _echo_loop(orc, n, delay_s, with_filter, burst)
    assert np.abs(o[:, :, 30 * RQ:]).max() > 1e-4  # (echoes)
    for i in range(g.shape[0]):
        for c in range(n):
            assert rms(g[i, c], o[i, c]) <= 1e-6, (i, c, plan)


@pytest.mark.gpu
@pytest.mark.parametrize("n,n_out", [(6, 2), (4, 2), (6, 1), (4, 1)])
def test_a_wide_signal_that_is_in_fact_mono_in_front_of_a_narrower_node(hip, orc, n, n_out):
    """wide fuzz seed 293 (round 6): a WaveShaper whose curve does not map 0 to 0 renders silence — ONE channel in the reference — into
    a signal, so behind a 5.1 source that starts late its output is mono at first: a stereo destination hears it on L and R as it is
    (mono -> stereo), not through the 5.1 -> stereo matrix the static plan applied to its six static channels (2.41 x).  A DOWN-mix of
    a producer wider than stereo is count-sensitive like every mix above stereo: the exact per-quantum counts of the dynamic plan"""
    outs = []
    for be in (hip, orc):
        ctx = waa.OfflineAudioContext(n_out, 16 * RQ, SR, n_instances=2, binding=be)
        src = ctx.create_buffer_source()
        src.set_buffer_batch(_noise(2, n, 16 * RQ, 61) * 0.3, SR)
        src.start_at(5.5 * RQ / SR)
        sh = ctx.create_wave_shaper()
        sh.set_curve((np.tanh(np.linspace(-2, 2, 65)) + 0.2).astype(np.float32))
        src.connect(sh).connect(ctx.destination())
        if be is hip:
            assert "dyn_kernel" in ctx.plan_describe()
        outs.append(ctx.start_rendering_sync().data)
        ctx.close()
    g, o = outs
    assert np.allclose(o[0, :, :5 * RQ], 0.2, atol=1e-6)  # curve(0), on every output channel, as it is
    for i in range(2):
        for c in range(n_out):
            assert rms(g[i, c], o[i, c]) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("n", [8, 16, 32])
def test_a_wide_source_that_ends_early_in_front_of_a_delay_and_a_filter(hip, orc, n):
    """the usual multi-channel case: a file shorter than the render in front of a DelayNode and a BiquadFilterNode.  When the source
    ends the DelayNode's input is silent = ONE channel: the reference re-mixes the line to mono (channels 1 ... n-1 of the delayed
    audio are gone, delay.rs:428-489), the filter keeps its n states and rings out.  Exact per-quantum counts: dyn_kernel<32>"""
    length = 40 * RQ
    outs = []
    for be in (hip, orc):
        ctx = waa.OfflineAudioContext(n, length, SR, n_instances=2, binding=be)
        src = ctx.create_buffer_source()
        src.set_buffer_batch(_noise(2, n, 12 * RQ + 50, 71), SR)
        src.start_at(3.3 * RQ / SR)
        d = ctx.create_delay(0.1)
        d.delay_time.set_value(0.02)
        bq = ctx.create_biquad_filter(type_="lowpass", frequency=1200.0, q=3.0)
        src.connect(d).connect(bq).connect(ctx.destination())
        src.connect(ctx.destination())
        if be is hip:
            assert "dyn_kernel" in ctx.plan_describe()
        outs.append(ctx.start_rendering_sync().data)
        ctx.close()
    g, o = outs
    assert np.abs(o[:, 0, 18 * RQ:24 * RQ]).max() > 1e-3 and np.all(o[:, 1:, 24 * RQ:] == 0)  # (channel 0 rings on; the others are gone)
    assert np.abs(o[:, 1:, 5 * RQ:14 * RQ]).max() > 0.1
    for i in range(2):
        for c in range(n):
            assert rms(g[i, c], o[i, c]) <= 1e-6, (i, c)
