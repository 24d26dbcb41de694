"""The AudioBufferSourceNode playhead replay (waa_schedule.cpp, restating audio_buffer_source.rs:422-845) takes steady-state
runs of frames without the `almost` snapping tests that cannot fire there (round 5: one 10 s slow-track replay was 9 ms of a
plan).  The tables must be BIT-identical to the frame-by-frame form: under WAA_SCHED_VERIFY=1 (measurement build) every
schedule is replayed both ways and compared — record by record, k as raw bits — and the library counts mismatches.  CPU only:
plan-only batches."""
import ctypes as C

import numpy as np
import pytest

import web_audio_api_rs_amd as waa

pytestmark = pytest.mark.measure
RQ = 128


def _counts(hip):
    fn = hip.lib.waa_debug_sched_verify
    fn.restype, fn.argtypes = C.c_uint64, [C.POINTER(C.c_uint64)]
    checked = C.c_uint64()
    bad = fn(C.byref(checked))
    return int(checked.value), int(bad)


def _plan(hip, length, sr, buf, buf_sr, rate, detune, loop, loop_start, loop_end, start, offset, duration, stop, n_inst=1):
    ctx = waa.OfflineAudioContext(1, length, sr, n_instances=n_inst, binding=hip, device=waa.PLAN_ONLY)
    src = ctx.create_buffer_source()
    src.set_buffer(waa.AudioBuffer(buf, buf_sr))
    src.playback_rate.set_value(rate)
    src.detune.set_value(detune)
    if loop:
        src.set_loop(True)
        if loop_start is not None:
            src.set_loop_start(loop_start)
            src.set_loop_end(loop_end)
    src.connect(ctx.destination())
    if duration is None:
        src.start_at_with_offset(start, offset)
    else:
        src.start_at_with_offset_and_duration(start, offset, duration)
    if stop is not None:
        src.stop_at(stop)
    text = ctx.plan_describe()
    ctx.close()
    return text


def test_c5_schedule_is_identical_with_and_without_runs(hip, monkeypatch):
    """BASELINE config 5's schedules: playbackRate 1.5 looping, and a 38 kHz buffer in a 48 kHz context, 10 s"""
    monkeypatch.setenv("WAA_SCHED_VERIFY", "1")
    c0, b0 = _counts(hip)
    buf = np.zeros((2, 65536), np.float32)
    for rate, bsr in ((1.5, 48000.0), (1.0, 38000.0)):
        text = _plan(hip, 480000, 48000.0, buf, bsr, rate, 0.0, True, None, None, 0.0, 0.0, None, None, n_inst=3)
        assert "slow=3750" in text
    c1, b1 = _counts(hip)
    assert c1 - c0 >= 2 and b1 == b0


def test_random_schedules_are_identical_with_and_without_runs(hip, monkeypatch):
    """rates of both signs, detune, loops with interior loop points (the playhead crosses them thousands of times), sub-sample
    starts, offsets, durations and stops that end inside a block, buffers at other rates"""
    monkeypatch.setenv("WAA_SCHED_VERIFY", "1")
    c0, b0 = _counts(hip)
    rng = np.random.default_rng(11)
    n = 0
    for case in range(400):
        sr = float(rng.choice([8000.0, 44100.0, 48000.0, 96000.0]))
        length = int(rng.integers(RQ * 3, RQ * 120))
        frames = int(rng.integers(2, 6000))
        buf_sr = float(rng.choice([sr, sr, 22050.0, 38000.0, 48000.0]))
        rate = float(rng.choice([1.0, 1.5, 0.5, 2.0, -1.0, -0.37, 0.999, 3.25, 1e-3]))
        detune = float(rng.choice([0.0, 0.0, 1200.0, -700.0, 33.0]))
        loop = bool(rng.random() < 0.6)
        dur_buf = frames / buf_sr
        ls, le = (None, None)
        if loop and rng.random() < 0.6:
            a, b = sorted(rng.uniform(0.0, dur_buf, 2))
            ls, le = float(a), float(b)
        start = float(rng.choice([0.0, 0.0, 1.0 / sr, 0.37 * RQ / sr, rng.uniform(0, length / sr)]))
        offset = float(rng.choice([0.0, 0.0, rng.uniform(0, dur_buf)]))
        duration = None if rng.random() < 0.6 else float(rng.uniform(0, 1.5 * length / sr))
        stop = None if rng.random() < 0.6 else float(rng.uniform(start, 1.2 * length / sr))
        buf = np.zeros((1, frames), np.float32)
        _plan(hip, length, sr, buf, buf_sr, rate, detune, loop, ls, le, start, offset, duration, stop)
        n += 1
    c1, b1 = _counts(hip)
    assert c1 - c0 >= n, (c0, c1, n)
    assert b1 == b0, f"{b1 - b0} of {c1 - c0} schedules differ between the two forms"


def test_runs_are_taken(hip, monkeypatch):
    """the fast form must not silently be the slow one: a 10 s slow-track replay renders (nearly) all of its frames in
    steady-state runs, and none with WAA_SCHED_NO_RUNS=1 (a frame counter of the library, not a stopwatch)"""
    fn = hip.lib.waa_debug_sched_run_frames
    fn.restype, fn.argtypes = C.c_uint64, []
    buf = np.zeros((2, 65536), np.float32)
    n0 = fn()
    _plan(hip, 480000, 48000.0, buf, 48000.0, 1.5, 0.0, True, None, None, 0.0, 0.0, None, None)
    n1 = fn()
    # (the silence replay and the source tables share ONE replay per plan: at least 95 % of its 480 000 frames in runs)
    assert n1 - n0 >= 0.95 * 480000, n1 - n0
    monkeypatch.setenv("WAA_SCHED_NO_RUNS", "1")
    _plan(hip, 480000, 48000.0, buf, 48000.0, 1.5, 0.0, True, None, None, 0.0, 0.0, None, None)
    assert fn() == n1
