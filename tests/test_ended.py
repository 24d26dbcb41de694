"""`ended` of scheduled sources: the reference's renderers call send_ended_event() in a
definite render quantum (audio_buffer_source.rs:446-461,834-842; constant_source.rs:204-262; oscillator.rs:382-465)
or, for sources still running at the end, from before_drop when the graph is unloaded (render/thread.rs:398-411).
The device engine renders node-major, so it reports per source and instance WHEN the event is due; a host shim fires
`onended` from that.  The oracle records the quantum while it renders quantum by quantum; the product derives it from
its host-side scheduling replay — no device needed (plan-only batches)."""
import numpy as np
import pytest

import web_audio_api_rs_amd as waa
from web_audio_api_rs_amd.api import ENDED_AT_UNLOAD, ENDED_NEVER

RQ = 128
SR = 48000.0


def build(be, n_inst=1, quanta=20, device=None):
    kw = {} if device is None else {"device": device}
    return waa.OfflineAudioContext(1, RQ * quanta, SR, n_instances=n_inst, binding=be, **kw)


def run(c, be_is_oracle):
    if be_is_oracle:
        c.start_rendering_sync()
    else:
        c.prepare()


@pytest.fixture(params=["orc", "hip-plan-only"])
def mk(request, orc, hip):
    """(binding, context kwargs): the oracle renders on the CPU, the product only plans (no GPU needed)"""
    if request.param == "orc":
        return orc, True, None
    return hip, False, waa.PLAN_ONLY


def ctx_for(mk, **kw):
    be, is_orc, device = mk
    return build(be, device=device, **kw), is_orc


def test_buffer_source_ends_when_the_buffer_is_exhausted(mk):
    c, is_orc = ctx_for(mk)
    src = c.create_buffer_source()
    src.set_buffer(waa.AudioBuffer(np.ones((1, RQ * 3 + 5), np.float32), SR))  # ends inside quantum 3
    src.connect(c.destination())
    src.start()
    run(c, is_orc)
    assert src.ended_quantum() == 3


def test_buffer_source_exact_multiple_of_the_quantum(mk):
    """buffer_time reaches buffer_duration at the end of quantum 2 (audio_buffer_source.rs:834-842)"""
    c, is_orc = ctx_for(mk)
    src = c.create_buffer_source()
    src.set_buffer(waa.AudioBuffer(np.ones((1, RQ * 3), np.float32), SR))
    src.connect(c.destination())
    src.start()
    run(c, is_orc)
    assert src.ended_quantum() == 2


def test_buffer_source_stop_time_and_late_start(mk):
    c, is_orc = ctx_for(mk)
    src = c.create_buffer_source()
    src.set_buffer(waa.AudioBuffer(np.ones((1, RQ * 50), np.float32), SR))
    src.connect(c.destination())
    src.start_at(RQ * 2.5 / SR)
    src.stop_at(RQ * 7.25 / SR)  # next_block_time >= stop_time first holds in quantum 7
    run(c, is_orc)
    assert src.ended_quantum() == 7


def test_looping_source_only_ends_at_unload(mk):
    c, is_orc = ctx_for(mk)
    src = c.create_buffer_source()
    src.set_buffer(waa.AudioBuffer(np.ones((1, 100), np.float32), SR))
    src.set_loop(True)
    src.connect(c.destination())
    src.start()
    run(c, is_orc)
    assert src.ended_quantum() == ENDED_AT_UNLOAD


def test_never_started_source_never_ends(mk):
    c, is_orc = ctx_for(mk)
    src = c.create_buffer_source()
    src.set_buffer(waa.AudioBuffer(np.ones((1, 100), np.float32), SR))
    src.connect(c.destination())
    osc = c.create_oscillator()
    osc.connect(c.destination())
    run(c, is_orc)
    assert src.ended_quantum() == ENDED_NEVER
    assert osc.ended_quantum() == ENDED_NEVER


def test_start_beyond_the_render_never_ends(mk):
    """before_drop fires only if current_time >= start_time or >= stop_time (audio_buffer_source.rs:872-878)"""
    c, is_orc = ctx_for(mk)
    k = c.create_constant_source()
    k.connect(c.destination())
    k.start_at(1.0)  # the render is 20 quanta = 53 ms long
    run(c, is_orc)
    assert k.ended_quantum() == ENDED_NEVER


def test_start_with_a_null_buffer_ends_immediately(mk):
    """audio_buffer_source.rs:443-451 (wpt audiobuffersource-start-null-buffer)"""
    c, is_orc = ctx_for(mk)
    src = c.create_buffer_source()
    src.connect(c.destination())
    src.start_at(RQ * 4 / SR)
    run(c, is_orc)
    assert src.ended_quantum() == 0


def test_constant_source_and_oscillator_stop(mk):
    c, is_orc = ctx_for(mk)
    k = c.create_constant_source()
    o = c.create_oscillator(frequency=440.0)
    running = c.create_oscillator(frequency=220.0)
    for n in (k, o, running):
        n.connect(c.destination())
    k.start()
    k.stop_at(RQ * 4.5 / SR)  # inside quantum 4: `still_running = stop > next_block_time` is false there
    o.start_at(RQ * 1.5 / SR)
    o.stop_at(RQ * 9.5 / SR)  # inside quantum 9
    running.start()
    run(c, is_orc)
    assert k.ended_quantum() == 4
    assert o.ended_quantum() == 9
    assert running.ended_quantum() == ENDED_AT_UNLOAD


def test_stop_before_start_ends_in_the_quantum_of_the_stop(mk):
    """constant_source.rs:204-212 / audio_buffer_source.rs:454-461: start beyond the block, stop inside it"""
    c, is_orc = ctx_for(mk)
    k = c.create_constant_source()
    k.connect(c.destination())
    k.start_at(RQ * 10 / SR)
    k.stop_at(RQ * 3.5 / SR)
    run(c, is_orc)
    assert k.ended_quantum() == 3


def test_per_instance_schedules(mk):
    be, is_orc, device = mk
    c = build(be, n_inst=3, device=device)
    src = c.create_buffer_source()
    src.set_buffer_batch(np.ones((3, 1, RQ * 6), np.float32), SR)
    src.connect(c.destination())
    for i, when in enumerate((0.0, RQ * 2 / SR, RQ * 30 / SR)):
        src.start_at(when, instance=i)
    run(c, is_orc)
    assert [src.ended_quantum(i) for i in range(3)] == [5, 7, ENDED_NEVER]


def test_playback_rate_changes_the_end(mk):
    c, is_orc = ctx_for(mk)
    src = c.create_buffer_source()
    src.set_buffer(waa.AudioBuffer(np.ones((1, RQ * 8 - 50), np.float32), SR))
    src.playback_rate.set_value(2.0)  # consumes 256 buffer frames per quantum: exhausted inside quantum 3
    src.connect(c.destination())
    src.start()
    run(c, is_orc)
    assert src.ended_quantum() == 3


def test_not_a_source_is_refused(mk):
    c, is_orc = ctx_for(mk)
    g = c.create_gain()
    g.connect(c.destination())
    run(c, is_orc)
    q = waa.api.C.c_int64()
    assert c._b.source_ended(c._handle, g.id, 0, waa.api.C.byref(q)) == 1  # WAA_ERR_INVALID_ARGUMENT


def test_block_boundary_stop_times_follow_the_f64_arithmetic(orc, hip):
    """stop times ON block boundaries: whether `stop <= next_block_time` holds in quantum q - 1 or only in q depends
    on the rounding of current_time + dt * 128 (thread.rs:360, constant_source.rs:201-202); the product's closed-form
    replay has to make the same call as the oracle's quantum-by-quantum render for every boundary"""
    quanta = 40
    results = []
    for be, device in ((orc, None), (hip, waa.PLAN_ONLY)):
        kw = {} if device is None else {"device": device}
        c = waa.OfflineAudioContext(1, RQ * quanta, 44100.0, n_instances=quanta, binding=be, **kw)
        nodes = [c.create_constant_source(), c.create_oscillator(), c.create_buffer_source()]
        nodes[2].set_buffer(waa.AudioBuffer(np.ones((1, RQ * 64), np.float32), 44100.0))
        for n in nodes:
            n.connect(c.destination())
            for i in range(quanta):
                n.start_at(0.0, instance=i)
                n.stop_at(RQ * (i + 1) / 44100.0, instance=i)  # exactly the end of quantum i
        if device is None:
            c.start_rendering_sync()
        else:
            c.prepare()
        results.append([[n.ended_quantum(i) for i in range(quanta)] for n in nodes])
    assert results[0] == results[1]
    # sanity: always quantum i or i + 1, never anything else
    for per_node in results[0]:
        assert all(q in (i, i + 1, ENDED_AT_UNLOAD) for i, q in enumerate(per_node))


# --------------------------------------------------------------------------- the reference's own onended tests
def _kat_ctx(mk, length=44100, sr=44100.0, channels=2):
    be, is_orc, device = mk
    kw = {} if device is None else {"device": device}
    return waa.OfflineAudioContext(channels, length, sr, binding=be, **kw), is_orc


def _make(c, kind):
    return {"constant": c.create_constant_source, "buffer": c.create_buffer_source, "oscillator": c.create_oscillator}[kind]()


@pytest.mark.parametrize("kind", ["constant", "buffer", "oscillator"])
def test_reference_ended_event_kats(mk, kind):
    """scheduled_source.rs:135-263 (run_ended_event, run_no_ended_event, run_exact_ended_event,
    run_implicit_ended_event), each for ConstantSource, AudioBufferSource (WITHOUT a buffer, as there) and Oscillator:
    the reference only asserts whether `onended` fired by the end of start_rendering_sync."""
    fired = {}
    for name, start, stop in (("ended", 0.0, 0.5), ("never started", None, None), ("exact", 0.0, 1.0), ("implicit", 0.0, None)):
        c, is_orc = _kat_ctx(mk)
        src = _make(c, kind)
        src.connect(c.destination())
        if start is not None:
            src.start_at(start)
        if stop is not None:
            src.stop_at(stop)
        run(c, is_orc)
        fired[name] = src.ended_quantum() != ENDED_NEVER
        c.close()
    assert fired == {"ended": True, "never started": False, "exact": True, "implicit": True}


def test_reference_onended_before_drop(mk):
    """audio_buffer_source.rs:2042-2071: the buffer is longer than the render, the event still fires (before_drop)"""
    c, is_orc = _kat_ctx(mk, length=RQ, sr=48000.0, channels=1)
    src = c.create_buffer_source()
    buf = np.zeros((1, RQ * 2), np.float32)
    buf[0, 0] = 1.0
    src.set_buffer(waa.AudioBuffer(buf, 48000.0))
    src.connect(c.destination())
    src.start()
    run(c, is_orc)
    assert src.ended_quantum() == ENDED_AT_UNLOAD


def test_reference_null_buffer_start_ends_before_start_time(mk):
    """audio_buffer_source.rs:1508-1534: start_at(0.75) with no buffer; `ended` is already set when the context is
    suspended at 0.5 s"""
    c, is_orc = _kat_ctx(mk, length=48000, sr=48000.0, channels=1)
    src = c.create_buffer_source()
    src.connect(c.destination())
    src.start_at(0.75)
    run(c, is_orc)
    q = src.ended_quantum()
    assert 0 <= q < int(0.5 * 48000.0) // RQ


def test_reference_osc_stop_before_start_ends_without_waiting(mk):
    """oscillator.rs:1248-1276: start two quanta ahead, stop() now: `ended` has fired when the context is suspended
    at the start of quantum 1"""
    c, is_orc = _kat_ctx(mk, length=RQ * 4, sr=44100.0, channels=1)
    osc = c.create_oscillator()
    osc.connect(c.destination())
    osc.start_at(2.0 * RQ / 44100.0)
    osc.stop()
    run(c, is_orc)
    assert osc.ended_quantum() == 0


@pytest.mark.parametrize("start_frame", [0, 1])
def test_reference_one_sample_buffer_ends_in_the_first_quantum(mk, start_frame):
    """audio_buffer_source.rs:1919-1983 (fast track / slow track): a one-sample buffer; the handler of `ended` runs
    before the second quantum is rendered (it switches looping on and the output must not restart)"""
    c, is_orc = _kat_ctx(mk, length=RQ * 4, sr=48000.0, channels=1)
    src = c.create_buffer_source()
    src.set_buffer(waa.AudioBuffer(np.ones((1, 1), np.float32), 48000.0))
    src.connect(c.destination())
    src.start_at(start_frame / 48000.0)
    run(c, is_orc)
    assert src.ended_quantum() == 0


# --------------------------------------------------------------------------- differential fuzz (CPU only)
def _random_source(be, seed, device=None):
    rng = np.random.default_rng(seed)
    kw = {} if device is None else {"device": device}
    quanta = int(rng.integers(5, 60))
    c = waa.OfflineAudioContext(1, RQ * quanta - int(rng.integers(0, 100)), SR, binding=be, **kw)
    src = c.create_buffer_source()
    frames = int(rng.choice([1, 2, 100, 128, 129, 1000, 4097, 6000]))
    bsr = float(rng.choice([48000.0, 44100.0, 22050.0, 96000.0]))
    src.set_buffer(waa.AudioBuffer(rng.uniform(-1, 1, (1, frames)).astype(np.float32), bsr))
    src.playback_rate.set_value(float(rng.choice([1.0, 1.0, 0.5, 1.5, 2.0, -1.0, -0.5, 0.0])))
    src.detune.set_value(float(rng.choice([0.0, 0.0, 1200.0, -1200.0, 100.0])))
    if rng.random() < 0.4:
        src.set_loop(True)
        if rng.random() < 0.7:
            src.set_loop_start(float(rng.uniform(0, frames / bsr)))
        if rng.random() < 0.7:
            src.set_loop_end(float(rng.uniform(0, frames / bsr * 1.2)))
    start = float(rng.choice([0.0, 0.0, 1 / SR, 128 / SR, 300.5 / SR, 0.02]))
    offset = float(rng.choice([0.0, 0.0, 10 / bsr, frames / bsr * 0.5, frames / bsr * 1.5]))
    duration = float(rng.choice([1.7976931348623157e308, 1.7976931348623157e308, 0.001, 0.01, 0.0]))
    src.start_at_with_offset_and_duration(start, offset, duration)
    if rng.random() < 0.4:
        src.stop_at(float(rng.choice([0.0, start, start + 0.0005, start + 0.01, 0.05])))
    src.connect(c.destination())
    return c, src


def test_random_schedules_end_in_the_same_quantum(orc, hip):
    """The product derives the `ended` quantum from its host-side scheduling replay (waa_schedule.cpp), the oracle
    from rendering the source quantum by quantum: 300 random AudioBufferSources — buffer lengths and sample rates,
    forward / reverse / zero playback rates, detune, loops with and without loop points, offsets past the end,
    durations, stop times before, at and after the start — have to agree.  (1500 seeds run once: no mismatch.)"""
    for seed in range(300):
        co, so = _random_source(orc, seed)
        co.start_rendering_sync()
        expect = so.ended_quantum()
        co.close()
        ch, sh = _random_source(hip, seed, waa.PLAN_ONLY)
        ch.prepare()
        assert sh.ended_quantum() == expect, seed
        ch.close()
