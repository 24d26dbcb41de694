"""DelayNode outside a feedback loop: the reference's own tests re-typed
(src/node/delay.rs:750-1206) on both backends, plus GPU-vs-oracle parity on seeded inputs.  The oracle keeps
the reference's ring of render quanta; the device path gathers from the node's input in absolute time —
the KATs below pin both."""
import numpy as np
import pytest

import web_audio_api_rs_amd as waa
from graphs import rms_err, white_noise

RQ = 128


def ctx(be, channels, length, sr, **kw):
    return waa.OfflineAudioContext(channels, length, sr, binding=be, **kw)


def dirac_through_delay(be, delay_frames, length, sr=48000.0, max_delay=2.0, start=0.0):
    c = ctx(be, 1, length, sr)
    delay = c.create_delay(max_delay)
    delay.delay_time.set_value(np.float32(delay_frames) / np.float32(sr))
    delay.connect(c.destination())
    src = c.create_buffer_source()
    src.connect(delay)
    src.set_buffer(waa.AudioBuffer(np.array([[1.0]], np.float32), sr))
    src.start_at(start)
    return c.start_rendering_sync().data[0, 0]


# --------------------------------------------------------------------------- reference KATs
def test_constructor_validation(be):
    """delay.rs:290-293"""
    c = ctx(be, 1, 128, 48000.0)
    for bad in (0.0, -1.0, 180.0, 200.0):
        with pytest.raises(waa.WaaError, match="NotSupportedError"):
            c.create_delay(bad)
    d = c.create_delay(1.0, delay_time=0.12)
    assert d.delay_time.value == pytest.approx(0.12)  # delay.rs:755-764


def test_c_abi_rejects_bad_max_delay(be):
    c = ctx(be, 1, 128, 48000.0)
    d = c.create_delay(1.0)
    d.max_delay_time = 500.0  # past the Python mirror: the C entry point must refuse
    d.connect(c.destination())
    with pytest.raises(waa.WaaError, match="NotSupportedError"):
        c.prepare()


@pytest.mark.parametrize("frames", [128.0, 131.0, 197.0])
def test_sample_accurate(be, frames):
    """delay.rs:766-792, abs_all <= 1e-5"""
    out = dirac_through_delay(be, frames, 256)
    expected = np.zeros(256, np.float32)
    expected[int(frames)] = 1.0
    assert np.max(np.abs(out - expected)) <= 1e-5


@pytest.mark.parametrize("frames,e128,e129", [(128.5, 0.5, 0.5), (128.8, 0.2, 0.8)])
def test_sub_sample_accurate(be, frames, e128, e129):
    """delay.rs:794-848"""
    out = dirac_through_delay(be, frames, 256)
    expected = np.zeros(256, np.float32)
    expected[128], expected[129] = e128, e129
    assert np.max(np.abs(out - expected)) <= 1e-5


def test_multichannel(be):
    """delay.rs:850-881"""
    sr = 48000.0
    c = ctx(be, 2, 256, sr)
    delay = c.create_delay(2.0)
    delay.delay_time.set_value(128.0 / sr)
    delay.connect(c.destination())
    buf = np.zeros((2, 256), np.float32)
    buf[0, 0] = 1.0
    buf[1, 1] = 1.0
    src = c.create_buffer_source()
    src.connect(delay)
    src.set_buffer(waa.AudioBuffer(buf, sr))
    src.start_at(0.0)
    out = c.start_rendering_sync().data[0]
    exp = np.zeros((2, 256), np.float32)
    exp[0, 128] = 1.0
    exp[1, 129] = 1.0
    assert np.max(np.abs(out - exp)) <= 1e-5


def test_input_number_of_channels_change(be):
    """delay.rs:883-924: a mono source, then a stereo one a quantum later; the delay line is up-mixed"""
    sr = 48000.0
    c = ctx(be, 2, 3 * 128, sr)
    delay = c.create_delay(2.0)
    delay.delay_time.set_value(128.0 / sr)
    delay.connect(c.destination())
    one = np.zeros((1, 128), np.float32)
    one[0, 0] = 1.0
    src1 = c.create_buffer_source()
    src1.connect(delay)
    src1.set_buffer(waa.AudioBuffer(one, sr))
    src1.start_at(0.0)
    two = np.zeros((2, 256), np.float32)
    two[0, 0] = 1.0
    two[1, 1] = 1.0
    src2 = c.create_buffer_source()
    src2.connect(delay)
    src2.set_buffer(waa.AudioBuffer(two, sr))
    src2.start_at(128.0 / sr)
    out = c.start_rendering_sync().data[0]
    exp = np.zeros((2, 384), np.float32)
    exp[0, 128] = exp[0, 256] = 1.0
    exp[1, 128] = exp[1, 257] = 1.0
    assert np.max(np.abs(out - exp)) <= 1e-5


def test_node_stays_alive_long_enough(be):
    """delay.rs:926-960: the source starts in the 4th quantum"""
    sr = 48000.0
    out = dirac_through_delay(be, 128.0, 5 * 128, max_delay=1.0, start=128.0 * 3.0 / sr)
    exp = np.zeros(5 * 128, np.float32)
    exp[4 * 128] = 1.0
    assert np.max(np.abs(out - exp)) <= 1e-5


@pytest.mark.parametrize("i", [0, 1, 2, 63, 64, 100, 126, 127])
def test_subquantum_delay(be, i):
    """delay.rs:962-988 (all of 0..128 there; a spread here)"""
    out = dirac_through_delay(be, float(i), 128, max_delay=1.0)
    exp = np.zeros(128, np.float32)
    exp[i] = 1.0
    assert np.max(np.abs(out - exp)) <= 1e-5


@pytest.mark.parametrize("seconds", [1.0, 1.5])
def test_max_delay(be, seconds):
    """delay.rs:1022-1074 (wpt delaynode-max-default-delay / -nondefault-delay): delay == maxDelay, exact copy.
    Shortened tone (0.5 s) and render (seconds + 1 s) to keep the CPU suite quick."""
    sr = 44100.0
    tone_len = int(0.5 * sr)
    length = int((seconds + 1.0) * sr)
    i = np.arange(tone_len, dtype=np.float32)
    tone = np.sin(np.float32(20.0) * np.float32(2.0) * np.float32(np.pi) * i / np.float32(sr)).astype(np.float32)
    c = ctx(be, 1, length, sr)
    src = c.create_buffer_source()
    src.set_buffer(waa.AudioBuffer(tone[None, :], sr))
    delay = c.create_delay(seconds)
    delay.delay_time.set_value(seconds)
    src.connect(delay)
    delay.connect(c.destination())
    src.start_at(0.0)
    out = c.start_rendering_sync().data[0, 0]
    d = int(seconds * sr)
    exp = np.zeros(length, np.float32)
    exp[d:d + tone_len] = tone
    assert np.array_equal(out, exp)


@pytest.mark.parametrize("quanta", [1, 2])
def test_max_delay_multiple_of_quantum_size(be, quanta):
    """delay.rs:1116-1172: delay == maxDelay == 1 or 2 render quanta"""
    sr = 48000.0
    out = dirac_through_delay(be, 128.0 * quanta, (quanta + 1) * 128, max_delay=128.0 * quanta / sr)
    exp = np.zeros((quanta + 1) * 128, np.float32)
    exp[128 * quanta] = 1.0
    assert np.max(np.abs(out - exp)) <= 1e-5


def test_subquantum_delay_dynamic_lifetime(be):
    """delay.rs:1174-1204: constant source for 120 frames, delayed by 64"""
    sr = 48000.0
    c = ctx(be, 1, 3 * 128, sr)
    delay = c.create_delay(1.0)
    delay.delay_time.set_value(np.float32(64.0) / np.float32(sr))
    delay.connect(c.destination())
    src = c.create_constant_source()
    src.connect(delay)
    src.start_at(0.0)
    src.stop_at(120.0 / sr)
    out = c.start_rendering_sync().data[0, 0]
    exp = np.zeros(3 * 128, np.float32)
    exp[64:64 + 120] = 1.0
    assert np.max(np.abs(out - exp)) <= 1e-5


def _plan_of_delay_graph(hip, a_rate=False, second_consumer=None):
    c = waa.OfflineAudioContext(2, RQ * 64, 48000.0, n_instances=4, binding=hip, device=waa.PLAN_ONLY)
    src = c.create_buffer_source()
    src.set_buffer_batch(white_noise(4, 2, RQ * 64), 48000.0)
    d = c.create_delay(0.5, delay_time=0.01)
    if a_rate:
        d.delay_time.set_block(0, np.full((64, RQ), 0.01, np.float32))
    src.connect(c.create_gain(gain=0.5)).connect(d).connect(c.create_gain(gain=2.0)).connect(c.destination())
    if second_consumer == "delay":
        d.connect(c.create_delay(0.5, delay_time=0.02)).connect(c.destination())
    if second_consumer == "conv":
        d.connect(c.create_convolver(buffer=waa.AudioBuffer(np.ones((1, 300), np.float32), 48000.0))).connect(c.destination())
    src.start()
    plan = c.plan_describe()
    c.close()
    return plan


def test_plan_constant_delay_is_folded_into_its_consumer(hip):
    """constant delayTime, consumed by a chain: no pass of its own, the consumer gathers from the delay line"""
    plan = _plan_of_delay_graph(hip)
    assert "read by its consumers from the delay line" in plan and "delayed:2ch" in plan and "ring=" not in plan


@pytest.mark.measure
def test_plan_delay_is_node_major(hip, monkeypatch):
    """per-frame delayTime, a node-major consumer, or the switch: the gather kernel (waa_delay.hip)"""
    plan = _plan_of_delay_graph(hip, a_rate=True)
    assert "delay node" in plan and "ring=189 quanta" in plan and "delayTime=a-rate" in plan
    plan = _plan_of_delay_graph(hip, second_consumer="conv")
    assert plan.count("ring=189 quanta") == 1 and "delayTime=const" in plan
    # (a DelayNode behind a DelayNode: the second one's line is a mix of the first one's line, both folded)
    plan = _plan_of_delay_graph(hip, second_consumer="delay")
    assert "ring=" not in plan and plan.count("read by its consumers from the delay line") == 2
    monkeypatch.setenv("WAA_NO_DELAY_FOLD", "1")
    plan = _plan_of_delay_graph(hip)
    assert "delay node" in plan and "ring=189 quanta" in plan and "delayTime=const" in plan


# --------------------------------------------------------------------------- GPU parity on seeded inputs
def _echo(binding, noise, delay_s, max_delay=1.0, blocks=None, length=None, sr=48000.0):
    """src -> [dry] + [delay -> gain 0.5] -> destination (a feed-forward echo)"""
    n_inst, n_ch, frames = noise.shape
    c = waa.OfflineAudioContext(2, length or frames, sr, n_instances=n_inst, binding=binding)
    src = c.create_buffer_source()
    src.set_buffer_batch(noise, sr)
    d = c.create_delay(max_delay)
    if blocks is None:
        for i in range(n_inst):
            d.delay_time.set_value(delay_s[i], instance=i)
    else:
        d.delay_time.set_block(0, blocks)
    wet = c.create_gain(gain=0.5)
    src.connect(c.destination())
    src.connect(d).connect(wet).connect(c.destination())
    src.start()
    out = c.start_rendering_sync().data
    c.close()
    return out


@pytest.mark.gpu
def test_delay_parity_constant_per_instance(hip, orc):
    n = 8
    noise = white_noise(n, 2, 2048 * 2 + 333, seed0=3)
    delays = np.float32([0.0, 1.0 / 48000.0, 0.5 / 48000.0, 127.0 / 48000.0, 128.0 / 48000.0, 0.0123, 0.05, 0.08])
    g = _echo(hip, noise, delays, max_delay=0.08)
    o = _echo(orc, noise, delays, max_delay=0.08)
    assert np.array_equal(g, o)


@pytest.mark.gpu
@pytest.mark.parametrize("rate", ["k", "a"])
def test_delay_parity_automated(hip, orc, rate):
    """k-rate: one delayTime per quantum (the len-1 path of delay.rs:560-590); a-rate: 128 values per quantum
    (a chorus-like sweep, delay.rs:591-606)"""
    n, frames = 3, 2048 * 2
    nq = frames // RQ
    noise = white_noise(n, 2, frames, seed0=8)
    if rate == "k":
        blocks = (0.002 + 0.0015 * np.sin(np.arange(nq) * 0.3)).astype(np.float32)
    else:
        t = np.arange(nq * RQ, dtype=np.float64) / 48000.0
        blocks = (0.003 + 0.002 * np.sin(2 * np.pi * 3.0 * t)).astype(np.float32).reshape(nq, RQ)
    g = _echo(hip, noise, None, max_delay=0.01, blocks=blocks)
    o = _echo(orc, noise, None, max_delay=0.01, blocks=blocks)
    assert rms_err(g, o).max() <= 1e-7
    assert np.abs(g - o).max() <= 1e-6


@pytest.mark.gpu
def test_delay_parity_mono_six_channel_inputs(hip, orc):
    """channel counts other than stereo, delay behind a biquad"""
    for nch in (1, 4):
        noise = white_noise(2, nch, 2048 + 100, seed0=nch)
        outs = []
        for be_ in (hip, orc):
            c = waa.OfflineAudioContext(nch, 2048 + 100, 48000.0, n_instances=2, binding=be_)
            src = c.create_buffer_source()
            src.set_buffer_batch(noise, 48000.0)
            d = c.create_delay(0.02, delay_time=0.0071, channel_count=nch, channel_count_mode="explicit",
                               channel_interpretation="discrete")
            src.connect(d).connect(c.destination())
            src.start()
            outs.append(c.start_rendering_sync().data)
            c.close()
        assert np.array_equal(*outs)


@pytest.mark.measure
@pytest.mark.gpu
@pytest.mark.parametrize("fold", [True, False])
def test_folded_delay_forms_are_bit_identical(hip, orc, fold, monkeypatch):
    """the gather inside the consumer's input stage (IN_DELAYED, source read in place) and the gather kernel behind a
    materialised source: one arithmetic — identical to each other and to the oracle, including a buffer that ends before
    the render does (zeros beyond the view), delays of 0 / sub-quantum / unaligned frames and a delay behind a delay"""
    if not fold:
        monkeypatch.setenv("WAA_NO_DELAY_FOLD", "1")
        monkeypatch.setenv("WAA_NO_SOURCE_VIEW", "1")
    n, frames, length = 6, RQ * 40, RQ * 64 + 57
    noise = white_noise(n, 1, frames, seed0=17)  # (mono: the end of the source changes no channel count, the plan stays static)
    delays = np.float32([0.0, 3.0 / 48000.0, 130.25 / 48000.0, 1001.0 / 48000.0, 0.0301, 0.0499])
    outs = []
    for be_ in (hip, orc):
        c = waa.OfflineAudioContext(2, length, 48000.0, n_instances=n, binding=be_)
        src = c.create_buffer_source()
        src.set_buffer_batch(noise, 48000.0)
        d1 = c.create_delay(0.05)
        d2 = c.create_delay(0.05, delay_time=0.0107)
        for i in range(n):
            d1.delay_time.set_value(float(delays[i]), instance=i)
        src.connect(c.destination())
        src.connect(d1).connect(c.create_gain(gain=0.5)).connect(c.destination())
        d1.connect(d2).connect(c.create_gain(gain=0.25)).connect(c.destination())
        src.start()
        if be_ is hip:
            plan = c.plan_describe()
            assert ("delayed:1ch" in plan) == fold and "dynamic-count" not in plan
        outs.append(c.start_rendering_sync().data)
        c.close()
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.measure
@pytest.mark.gpu
def test_folded_delay_in_a_block_scheduled_loop(hip, orc, monkeypatch):
    """echo loop Delay <-> Gain with a Biquad tap: one launch per block with the folds, three without — same samples"""
    n, frames = 3, 2048 * 12
    noise = white_noise(n, 2, frames, seed0=23)

    def render(be_):
        c = waa.OfflineAudioContext(2, frames, 48000.0, n_instances=n, binding=be_)
        src = c.create_buffer_source()
        src.set_buffer_batch(noise, 48000.0)  # (as long as the render: no channel-count change, the plan stays static)
        d = c.create_delay(1.0, delay_time=0.1)  # 4800 frames: blocks of 2 tiles
        g = c.create_gain(gain=0.6)
        src.connect(d)
        d.connect(g).connect(d)
        d.connect(c.create_biquad_filter(type_="highpass", frequency=300.0)).connect(c.destination())
        src.connect(c.destination())
        src.start()
        plan = c.plan_describe() if be_ is hip else ""
        out = c.start_rendering_sync().data
        c.close()
        return out, plan

    folded, plan = render(hip)
    assert "1 step(s) per block" in plan
    monkeypatch.setenv("WAA_NO_LOOP_FOLD", "1")
    plain, plan = render(hip)
    assert "3 step(s) per block" in plan
    ref, _ = render(orc)
    assert np.array_equal(folded, plain)
    assert rms_err(folded, ref).max() <= 1e-7


def _ff_echo(binding, noise, delays, variant, length=None, plan_only=False):
    """the feed-forward echo family the LDS-ring kernel renders with nothing fed back (waa_echo.hip, echo_feed_forward):
    dry+wet  src -> destination, src -> Delay -> Gain(0.5) -> destination
    wet+dry  the same, connected in the other order (the other summation order)
    wet      src -> Delay -> destination only
    other    the wet path plus ANOTHER source into the destination (stays on the tile-parallel kernel)"""
    n_inst, n_ch, frames = noise.shape
    c = waa.OfflineAudioContext(2, length or frames, 48000.0, n_instances=n_inst, binding=binding,
                                **({"device": waa.PLAN_ONLY} if plan_only else {}))
    src = c.create_buffer_source()
    src.set_buffer_batch(noise, 48000.0)
    d = c.create_delay(0.4)
    for i in range(n_inst):
        d.delay_time.set_value(delays[i], instance=i)
    if variant == "dry+wet":
        src.connect(c.destination())
    if variant == "wet":
        src.connect(d).connect(c.destination())
    else:
        src.connect(d).connect(c.create_gain(gain=0.5)).connect(c.destination())
    if variant == "wet+dry":
        src.connect(c.destination())
    if variant == "other":
        o2 = c.create_buffer_source()
        o2.set_buffer_batch(noise[:, :, ::-1].copy(), 48000.0)
        o2.connect(c.destination())
        o2.start()
    src.start()
    plan = c.plan_describe() if binding.prefix == "waa_" else ""
    out = None if plan_only else c.start_rendering_sync().data
    c.close()
    return out, plan


FF_DELAYS = (np.float64([1032, 1033.5, 3000.25, 4800, 9000.75, 15352]) / 48000.0).astype(np.float32)


@pytest.mark.measure
@pytest.mark.parametrize("variant,ring", [("dry+wet", True), ("wet+dry", True), ("wet", True), ("other", False)])
def test_plan_feed_forward_echo_out_of_the_ring(hip, variant, ring, monkeypatch):
    noise = white_noise(6, 2, 2048 * 4, seed0=5)
    assert "LDS-ring" not in _ff_echo(hip, noise, FF_DELAYS, variant, plan_only=True)[1]   # (6 instances: below one per CU)
    monkeypatch.setenv("WAA_ECHO_FF_MIN_INST", "1")
    plan = _ff_echo(hip, noise, FF_DELAYS, variant, plan_only=True)[1]
    assert ("LDS-ring kernel with nothing fed back: delay 1032 .. 15352 frames, chunks of 1024 frames" in plan) == ring
    if not ring:
        assert "sums a delayed signal but keeps the tile-parallel kernel: it " in plan   # (... has an input that is not X)
    late = FF_DELAYS.copy()
    late[0] = np.float32(1031.0 / 48000.0)
    assert "keeps the tile-parallel kernel: a delay outside the ring's window" in _ff_echo(hip, noise, late, "wet", plan_only=True)[1]
    monkeypatch.setenv("WAA_NO_ECHO_FF", "1")
    assert "LDS-ring" not in _ff_echo(hip, noise, FF_DELAYS, variant, plan_only=True)[1]


def test_plan_feed_forward_echo_takes_the_ring_from_one_instance_per_cu(hip):
    """the default threshold: 256 instances (one per CU of an MI355X) and the echo goes through the ring kernel, 255 and it
    keeps the tile-parallel launch"""
    for n, ring in ((255, False), (256, True)):
        noise = np.zeros((n, 1, 2048 * 2), np.float32)
        delays = np.full(n, 4800.0 / 48000.0, np.float32)
        plan = _ff_echo(hip, noise, delays, "dry+wet", plan_only=True)[1]
        assert ("LDS-ring kernel with nothing fed back" in plan) == ring, n


@pytest.mark.measure
@pytest.mark.gpu
@pytest.mark.parametrize("channels", [1, 2])
@pytest.mark.parametrize("variant", ["dry+wet", "wet+dry", "wet", "other"])
def test_parity_feed_forward_echo_out_of_the_ring(hip, orc, variant, channels, monkeypatch):
    """bit-identical to the tile-parallel launch and to the oracle: delays at both ends of the window, a ragged length,
    a mono source up-mixed by the destination, and a source that ends before the render does (zeros beyond its view)"""
    n, frames, length = 6, 2048 * 9 + 100, 2048 * 11 + 77
    noise = white_noise(n, channels, frames if channels == 1 else length, seed0=41)
    monkeypatch.setenv("WAA_ECHO_FF_MIN_INST", "1")
    ring, plan = _ff_echo(hip, noise, FF_DELAYS, variant, length=length)
    assert ("LDS-ring kernel with nothing fed back" in plan) == (variant != "other")
    o, _ = _ff_echo(orc, noise, FF_DELAYS, variant, length=length)
    assert np.array_equal(ring, o)
    monkeypatch.setenv("WAA_NO_ECHO_FF", "1")
    plain, plan = _ff_echo(hip, noise, FF_DELAYS, variant, length=length)
    assert "LDS-ring" not in plan and np.array_equal(plain, ring)
