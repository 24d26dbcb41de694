"""The reference's own known-answer tests, re-typed, run against BOTH implementations.

Every test is a unit/integration test of the reference (file:line in the docstring) driven
through the C ABI with the host mirror:

* backend ``orc``  — the CPU oracle (prefix orc_): this is what PINS the oracle; runs in the
  CPU suite (``-m "not gpu"``).
* backend ``hip``  — the product library on a real MI355X (prefix waa_): ``-m gpu``.

Tests that take ``orc``/``orc_lib`` directly exercise oracle-only helpers (CPU only).
"""
import json
import math
import os

import numpy as np
import pytest

import web_audio_api_rs_amd as waa

RQ = 128
HERE = os.path.dirname(os.path.abspath(__file__))
F32PI = np.float32(np.pi)


def ctx(be, channels, length, sr, **kw):
    return waa.OfflineAudioContext(channels, length, sr, binding=be, **kw)


def buf(data, sr):
    return waa.AudioBuffer(np.asarray(data, np.float32), sr)


# ----------------------------------------------------------------------------- biquad
KATS = json.load(open(os.path.join(HERE, "golden", "biquad_frequency_response.json")))


@pytest.mark.parametrize("name", sorted(KATS))
def test_biquad_frequency_responses(be, name):
    """src/node/biquad_filter.rs:1000-1412 — Chrome/Firefox vectors, abs_all <= 1e-6."""
    k = KATS[name]
    c = ctx(be, 1, 128, k["sample_rate"])
    f = c.create_biquad_filter()
    f.set_type(name)
    f.frequency.set_value(k["frequency"])
    f.q.set_value(k["q"])
    f.gain.set_value(k["gain"])
    mags, phases = f.get_frequency_response(k["freqs"])
    assert np.max(np.abs(mags - np.float32(k["expected_mags"]))) <= 1e-6
    assert np.max(np.abs(phases - np.float32(k["expected_phases"]))) <= 1e-6


def test_biquad_frequency_response_nan_outside_range(be):
    """src/node/biquad_filter.rs:1415-1436"""
    c = ctx(be, 1, 128, 44100.0)
    f = c.create_biquad_filter()
    mags, phases = f.get_frequency_response([-1.0, 22051.0])
    assert np.all(np.isnan(mags)) and np.all(np.isnan(phases))


def test_computed_freq(orc_lib):
    """src/node/biquad_filter.rs:920-931"""
    import ctypes as C
    orc_lib.orc_get_computed_freq.restype = C.c_float
    orc_lib.orc_get_computed_freq.argtypes = [C.c_float, C.c_float]
    assert orc_lib.orc_get_computed_freq(440.0, 0.0) == 440.0
    assert abs(orc_lib.orc_get_computed_freq(440.0, 1200.0) - 880.0) <= 1e-4
    assert abs(orc_lib.orc_get_computed_freq(440.0, -1200.0) - 220.0) <= 1e-4


def test_biquad_render_matches_lfilter(be, orc):
    """The reference pins no biquad *samples* (SURVEY §8c); cross-check the restated
    DF-I recurrence against scipy.signal.lfilter in f64 with the restated coefficients."""
    import ctypes as C
    from scipy.signal import lfilter
    sr, n = 48000.0, 128 * 40
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, (2, n)).astype(np.float32)
    for ftype, f0, q, g in [("lowpass", 200.0, 1.0, 0.0), ("peaking", 3000.0, 2.0, 6.0), ("highshelf", 5000.0, 1.0, -9.0),
                            ("notch", 1000.0, 10.0, 0.0)]:
        c = ctx(be, 2, n, sr)
        src = c.create_buffer_source()
        src.set_buffer(buf(x, sr))
        flt = c.create_biquad_filter(type_=ftype, frequency=f0, q=q, gain=g)
        src.connect(flt).connect(c.destination())
        src.start()
        out = c.start_rendering_sync().data[0]
        co = (C.c_double * 5)()
        orc.lib.orc_biquad_coefs.argtypes = [C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                             C.POINTER(C.c_double)]
        orc.lib.orc_biquad_coefs(waa.BIQUAD_TYPE[ftype], sr, f0, 0.0, q, g, co)
        b, a = [co[0], co[1], co[2]], [1.0, co[3], co[4]]
        ref = lfilter(b, a, x.astype(np.float64), axis=1)
        assert np.max(np.abs(out - ref.astype(np.float32))) <= 1e-6


# ----------------------------------------------------------------------------- mixing
def _mix(orc_lib, data, to, interp=0):
    import ctypes as C
    data = np.ascontiguousarray(data, np.float32)
    out = np.zeros((to, RQ), np.float32)
    orc_lib.orc_mix.argtypes = [C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.c_int32, C.POINTER(C.c_float)]
    orc_lib.orc_mix(data.ctypes.data_as(C.POINTER(C.c_float)), data.shape[0], to, interp,
                    out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def test_mix_rules(orc_lib):
    """src/render/quantum.rs:285-505 (tests :803-1440): speaker up/down-mix matrices."""
    ch = [np.full(RQ, v, np.float32) for v in (1.0, 0.5, 0.25, 0.125, 0.75, 0.3)]
    # up-mix
    o = _mix(orc_lib, ch[:1], 2)
    assert np.all(o[0] == 1.0) and np.all(o[1] == 1.0)
    o = _mix(orc_lib, ch[:1], 4)
    assert np.all(o[0] == 1.0) and np.all(o[1] == 1.0) and np.all(o[2:] == 0.0)
    o = _mix(orc_lib, ch[:1], 6)
    assert np.all(o[2] == 1.0) and np.all(o[[0, 1, 3, 4, 5]] == 0.0)
    o = _mix(orc_lib, ch[:2], 4)
    assert np.all(o[0] == 1.0) and np.all(o[1] == 0.5) and np.all(o[2:] == 0.0)
    o = _mix(orc_lib, ch[:2], 6)
    assert np.all(o[0] == 1.0) and np.all(o[1] == 0.5) and np.all(o[2:] == 0.0)
    o = _mix(orc_lib, ch[:4], 6)
    assert np.all(o[0] == 1.0) and np.all(o[1] == 0.5) and np.all(o[2:4] == 0.0) and np.all(o[4] == 0.25) and np.all(
        o[5] == 0.125)
    # down-mix
    assert np.all(_mix(orc_lib, ch[:2], 1)[0] == np.float32(0.5) * (np.float32(1.0) + np.float32(0.5)))
    assert np.all(_mix(orc_lib, ch[:4], 1)[0] == np.float32(0.25) * np.float32(1.0 + 0.5 + 0.25 + 0.125))
    s = np.sqrt(np.float32(0.5))
    o = _mix(orc_lib, ch[:6], 1)
    expect = s * np.float32(1.5) + (np.float32(0.5) * np.float32(0.75 + 0.3) + np.float32(0.25))
    assert np.allclose(o[0], expect, atol=1e-7)
    o = _mix(orc_lib, ch[:4], 2)
    assert np.all(o[0] == np.float32(0.5) * np.float32(1.25)) and np.all(o[1] == np.float32(0.5) * np.float32(0.625))
    o = _mix(orc_lib, ch[:6], 2)
    assert np.allclose(o[0], 1.0 + s * (0.25 + 0.75), atol=1e-7) and np.allclose(o[1], 0.5 + s * (0.25 + 0.3), atol=1e-7)
    o = _mix(orc_lib, ch[:6], 4)
    assert np.allclose(o[0], 1.0 + s * 0.25, atol=1e-7) and np.allclose(o[1], 0.5 + s * 0.25, atol=1e-7)
    assert np.all(o[2] == 0.75) and np.all(o[3] == np.float32(0.3))
    # discrete: pad / truncate
    o = _mix(orc_lib, ch[:1], 2, interp=1)
    assert np.all(o[0] == 1.0) and np.all(o[1] == 0.0)
    o = _mix(orc_lib, ch[:2], 1, interp=1)
    assert np.all(o[0] == 1.0)


def _run_mixing(be, n_out, dest_interp, count, mode, interp):
    """tests/mixing.rs:9-37"""
    c = ctx(be, n_out, 128, 44100.0)
    c.destination().set_channel_interpretation(dest_interp)
    k = c.create_constant_source()
    k.start()
    g = c.create_gain()
    g.set_channel_count(count)
    g.set_channel_count_mode(mode)
    g.set_channel_interpretation(interp)
    k.connect(g).connect(c.destination())
    return c.start_rendering_sync().data[0]


def test_mixing_integration(be):
    """tests/mixing.rs:40-100"""
    ones, zeros = np.ones(128, np.float32), np.zeros(128, np.float32)
    o = _run_mixing(be, 1, "speakers", 1, "max", "speakers")
    assert np.array_equal(o[0], ones)
    o = _run_mixing(be, 2, "speakers", 2, "max", "speakers")
    assert np.array_equal(o[0], ones) and np.array_equal(o[1], ones)
    o = _run_mixing(be, 2, "discrete", 1, "max", "speakers")
    assert np.array_equal(o[0], ones) and np.array_equal(o[1], zeros)
    o = _run_mixing(be, 2, "discrete", 2, "max", "speakers")
    assert np.array_equal(o[0], ones) and np.array_equal(o[1], zeros)
    o = _run_mixing(be, 1, "discrete", 2, "max", "speakers")
    assert np.array_equal(o[0], ones)


def test_mixing_integration_quad(be):
    """tests/mixing.rs:56-65 (quad destination)."""
    ones, zeros = np.ones(128, np.float32), np.zeros(128, np.float32)
    o = _run_mixing(be, 4, "speakers", 4, "max", "speakers")
    assert np.array_equal(o[0], ones) and np.array_equal(o[1], ones) and np.array_equal(o[2], zeros) and np.array_equal(
        o[3], zeros)


def test_wide_channel_mixing_chain(be, orc):
    """quantum.rs:285-505 end to end: stereo source -> gain(6, explicit) -> gain(4, explicit) -> 5.1 destination
    and a 6 -> 1 down-mix, against the oracle (the oracle's matrices are pinned by test_mix_rules)."""
    sr = 44100.0
    rng = np.random.default_rng(2)
    data = rng.uniform(-1, 1, (2, 128 * 6)).astype(np.float32)
    outs = []
    for b in (be, orc):
        for n_out in (6, 1):
            c = ctx(b, n_out, 128 * 6, sr)
            s = c.create_buffer_source()
            s.set_buffer(buf(data, sr))
            g6 = c.create_gain(gain=0.5, channel_count=6, channel_count_mode="explicit", channel_interpretation="speakers")
            g4 = c.create_gain(gain=0.8, channel_count=4, channel_count_mode="explicit", channel_interpretation="speakers")
            s.connect(g6).connect(g4).connect(c.destination())
            s.start()
            outs.append(c.start_rendering_sync().data)
    assert np.array_equal(outs[0], outs[2]) and np.array_equal(outs[1], outs[3])


def test_offline_render_summing_and_truncation(be):
    """tests/offline.rs:11-46 — fan-in summing, non-multiple-of-128 length, mono->stereo."""
    length = 555
    c = ctx(be, 2, length, 44100.0)
    k1 = c.create_constant_source()
    k1.offset.set_value(2.0)
    k1.connect(c.destination())
    k2 = c.create_constant_source()
    k2.offset.set_value(-4.0)
    k2.connect(c.destination())
    k1.start()
    k2.start()
    out = c.start_rendering_sync()
    assert out.number_of_channels == 2 and out.length == length
    assert np.array_equal(out.data[0, 0], np.full(length, -2.0, np.float32))
    assert np.array_equal(out.data[0, 1], np.full(length, -2.0, np.float32))


def test_flush_denormals(be):
    """tests/denormals.rs:5-30"""
    c = ctx(be, 1, 128, 48000.0)
    s = c.create_constant_source()
    s.start()
    g1 = c.create_gain(gain=0.001)
    g2 = c.create_gain(gain=float(np.finfo(np.float32).tiny))
    g3 = c.create_gain(gain=float(np.finfo(np.float32).max))
    s.connect(g1).connect(g2).connect(g3).connect(c.destination())
    out = c.start_rendering_sync().data[0, 0]
    assert np.array_equal(out, np.zeros(128, np.float32))


def test_start_rendering_twice(be):
    """src/context/offline.rs:163 InvalidStateError"""
    c = ctx(be, 1, 128, 48000.0)
    c.start_rendering_sync()
    with pytest.raises(waa.WaaError, match="InvalidStateError"):
        c.start_rendering_sync()


# ----------------------------------------------------------------------------- stereo panner
def _pan(be, data, pan, **cfg):
    c = ctx(be, 2, 128, 44100.0)
    p = c.create_stereo_panner(pan=pan, **cfg)
    p.connect(c.destination())
    s = c.create_buffer_source()
    s.connect(p)
    s.set_buffer(buf(data, 44100.0))
    s.start()
    return c.start_rendering_sync().data[0]


def test_stereo_panner_mono(be):
    """src/node/stereo_panner.rs:370-462"""
    one = np.ones((1, 128), np.float32)
    cfg = dict(channel_count=1, channel_count_mode="clamped-max")
    o = _pan(be, one, -1.0, **cfg)
    assert np.array_equal(o[0], one[0]) and np.array_equal(o[1], np.zeros(128, np.float32))
    o = _pan(be, one, 1.0, **cfg)
    assert np.max(np.abs(o[0])) <= 1e-7 and np.array_equal(o[1], one[0])
    o = _pan(be, one, 0.0, **cfg)
    assert np.max(np.abs(o[0] * o[0] + o[1] * o[1] - 1.0)) <= 1.2e-7


def test_stereo_panner_stereo(be):
    """src/node/stereo_panner.rs:465-552"""
    ones = np.ones((2, 128), np.float32)
    o = _pan(be, ones, -1.0)
    assert np.array_equal(o[0], np.full(128, 2.0, np.float32)) and np.array_equal(o[1], np.zeros(128, np.float32))
    o = _pan(be, ones, 1.0)
    assert np.max(np.abs(o[0])) <= 1e-7 and np.array_equal(o[1], np.full(128, 2.0, np.float32))
    o = _pan(be, ones, 0.0)
    assert np.max(np.abs(o[0] - 1.0)) <= 1e-7 and np.array_equal(o[1], np.ones(128, np.float32))


def test_stereo_panner_rejects_max_mode(be):
    """src/node/stereo_panner.rs:60-68"""
    c = ctx(be, 2, 128, 44100.0)
    c.create_stereo_panner(channel_count=2, channel_count_mode="max")
    with pytest.raises(waa.WaaError, match="NotSupportedError"):
        c.start_rendering_sync()


# ----------------------------------------------------------------------------- panner / spatial
def test_spatial_geometry(orc_lib):
    """src/spatial.rs:313-394"""
    import ctypes as C
    F3 = C.c_float * 3
    orc_lib.orc_azimuth_elevation.argtypes = [F3, F3, F3, F3, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    orc_lib.orc_spatial_angle.argtypes = [F3, F3, F3]
    orc_lib.orc_spatial_angle.restype = C.c_float
    LP, LF, LU = F3(0, 0, 0), F3(0, 0, -1), F3(0, 1, 0)

    def azel(pos):
        az, el = C.c_float(), C.c_float()
        orc_lib.orc_azimuth_elevation(F3(*pos), LP, LF, LU, C.byref(az), C.byref(el))
        return az.value, el.value

    assert azel([0, 0, 0]) == (0.0, 0.0)
    for pos, (eaz, eel) in [([10, 0, 0], (90, 0)), ([-10, 0, 0], (-90, 0)), ([10, 0, -10], (45, 0)),
                            ([-10, 0, -10], (-45, 0))]:
        az, el = azel(pos)
        assert abs(az - eaz) <= 1e-3 and el == eel
    az, el = azel([0, -10, 0])
    assert abs(az) <= 1e-3 and abs(el + 90) <= 1e-3
    az, el = azel([0, 10, 0])
    assert abs(az) <= 1e-3 and abs(el - 90) <= 1e-3
    assert orc_lib.orc_spatial_angle(F3(0, 0, 0), F3(1, 0, 0), LP) == 0.0
    assert orc_lib.orc_spatial_angle(F3(10, 0, 0), F3(0, 0, 0), LP) == 0.0
    assert orc_lib.orc_spatial_angle(F3(1, 0, 0), F3(0, 1, 0), LP) == 90.0
    assert orc_lib.orc_spatial_angle(F3(1, 0, 0), F3(0, -1, 0), LP) == 90.0


def test_panner_equal_power_mono_to_stereo(be):
    """src/node/panner.rs:1081-1131"""
    sr = 44100.0
    c = ctx(be, 2, RQ * 4, sr)
    s = c.create_buffer_source()
    s.set_buffer(buf(np.ones((1, RQ)), sr))
    s.start()
    p = c.create_panner(panning_model="equalpower")
    p.set_channel_count(1)
    p.set_channel_count_mode("clamped-max")
    p.position_x.set_value(1.0)
    s.connect(p).connect(c.destination())
    o = c.start_rendering_sync().data[0]
    assert np.max(np.abs(o[0, :128])) <= 1e-6 and np.max(np.abs(o[1, :128] - 1.0)) <= 1e-6
    assert np.max(np.abs(o[:, 128:256])) <= 1e-6


def test_panner_equal_power_azimuth(be):
    """src/node/panner.rs:1133-1168"""
    sr = 44100.0
    c = ctx(be, 2, RQ, sr)
    s = c.create_buffer_source()
    s.set_buffer(buf(np.ones((1, RQ)), sr))
    s.start()
    p = c.create_panner()
    p.position_y.set_value(1.0)
    s.connect(p).connect(c.destination())
    o = c.start_rendering_sync().data[0]
    r = np.sqrt(np.float32(0.5))
    assert np.max(np.abs(o[0] - r)) <= 1e-6 and np.max(np.abs(o[1] - r)) <= 1e-6


def test_panner_equal_power_stereo_to_stereo(be):
    """src/node/panner.rs:1170-1223"""
    sr = 44100.0
    c = ctx(be, 2, RQ, sr)
    c.listener().set_position(10.0, 0.0, 0.0)
    c.listener().set_orientation(1.0, 0.0, 0.0, 0.0, 0.0, 1.0)
    s = c.create_buffer_source()
    s.set_buffer(buf(np.ones((2, RQ)), sr))
    s.start()
    p = c.create_panner()
    p.set_position(10.0, 10.0, 0.0)
    s.connect(p).connect(c.destination())
    o = c.start_rendering_sync().data[0]
    assert np.max(np.abs(o[0] - 0.2)) <= 1e-3 and np.max(np.abs(o[1])) <= 1e-3


def test_panner_hrtf_in_a_feedback_loop(be, orc):
    """HRTF panning inside a feedback loop: refused with status 4 until round 4; since round 5 the loop is rendered quantum block
    by quantum block around the node (tests/test_frozen_loops.py) — here the smallest such graph, on both backends, the device
    against the oracle"""
    def render(b):
        c = ctx(b, 2, 8 * RQ, 44100.0)
        s = c.create_constant_source()
        s.start()
        p = c.create_panner(panning_model="HRTF")
        d = c.create_delay(1.0, delay_time=0.01)
        s.connect(p).connect(d).connect(p)
        d.connect(c.destination())
        return c.start_rendering_sync().data
    out = render(be)
    assert np.isfinite(out).all() and float(np.abs(out).max()) > 1e-3
    if be.prefix != "orc_":
        ref = render(orc)
        assert np.sqrt(np.mean((out.astype(np.float64) - ref) ** 2, axis=-1)).max() <= 1e-6 * max(1.0, float(np.abs(ref).max()))


# ----------------------------------------------------------------------------- buffer source
def _play(be, data, sr, length=RQ, channels=1, buf_sr=None, setup=None):
    c = ctx(be, channels, length, sr)
    s = c.create_buffer_source()
    s.connect(c.destination())
    b = buf(data, buf_sr or sr)
    s.set_buffer(b)
    setup(s, b)
    return c.start_rendering_sync().data[0]


def test_source_sub_quantum_start(be):
    """src/node/audio_buffer_source.rs:974-995, :1036-1057"""
    sr = 48000.0
    o = _play(be, [[1.0]], sr, setup=lambda s, b: s.start_at(1.0 / sr))
    e = np.zeros(RQ, np.float32)
    e[1] = 1.0
    assert np.array_equal(o[0], e)
    o = _play(be, [[1.0]], sr, setup=lambda s, b: s.start_at(1.5 / sr))
    e = np.zeros(RQ, np.float32)
    e[2] = 0.5
    assert np.array_equal(o[0], e)


def test_source_sample_accurate_scheduling(be):
    """src/node/audio_buffer_source.rs:997-1033"""
    sr = 44100.0
    c = ctx(be, 2, int(4 * sr), sr)
    d = np.zeros((2, 512), np.float32)
    d[:, 0] = 1.0
    offsets = [0, 3, 512, 517, 1000, 1005, 20000, 21234, 37590]
    for idx in offsets:
        s = c.create_buffer_source()
        s.set_buffer(buf(d, sr))
        s.connect(c.destination())
        s.start_at(idx / sr)
    o = c.start_rendering_sync().data[0]
    assert np.array_equal(o[0], o[1])
    for idx in offsets:
        assert o[0, idx] != 0.0


def test_source_stop(be):
    """src/node/audio_buffer_source.rs:1059-1149"""
    sr = 48000.0
    z = np.zeros(RQ, np.float32)

    def pad(v):
        a = np.zeros((1, RQ), np.float32)
        a[0, :len(v)] = v
        return a

    o = _play(be, pad([0, 0, 0, 0, 1]), sr, setup=lambda s, b: (s.start_at(0.0), s.stop_at(4.0 / sr)))
    assert np.array_equal(o[0], z)
    o = _play(be, pad([0, 0, 0, 1]), sr, setup=lambda s, b: (s.start_at(1.0 / sr), s.stop_at(4.0 / sr)))
    assert np.array_equal(o[0], z)
    o = _play(be, pad([0, 0, 0, 0, 1, 1]), sr, setup=lambda s, b: (s.start_at(0.0), s.stop_at(4.5 / sr)))
    e = z.copy()
    e[4] = 1.0
    assert np.array_equal(o[0], e)
    o = _play(be, pad([0, 0, 0, 0, 1, 1]), sr, setup=lambda s, b: (s.start_at(1.0 / sr), s.stop_at(5.5 / sr)))
    e = z.copy()
    e[5] = 1.0
    assert np.array_equal(o[0], e)


@pytest.mark.parametrize("buf_sr", [22500, 38000, 43800, 48000, 96000])
def test_source_buffer_resampling(be, buf_sr):
    """src/node/audio_buffer_source.rs:1175-1217 — 1 Hz sine at 5 buffer rates, 1e-6."""
    base = 44100
    i = np.arange(buf_sr, dtype=np.float32)
    sine = np.sin(np.float32(1.0) * i / np.float32(buf_sr) * np.float32(2.0) * F32PI).astype(np.float32)
    o = _play(be, sine[None, :], float(base), length=base, buf_sr=float(buf_sr), setup=lambda s, b: s.start_at(0.0))
    j = np.arange(base, dtype=np.float32)
    exp = np.sin(j / np.float32(base) * np.float32(2.0) * F32PI).astype(np.float32)
    assert np.max(np.abs(o[0] - exp)) <= 1e-6


def _sine(n):
    i = np.arange(n, dtype=np.float32)
    return np.sin(i / np.float32(n) * np.float32(2.0) * F32PI).astype(np.float32)


def test_source_playback_rate_and_detune(be):
    """src/node/audio_buffer_source.rs:1220-1256, :1294-1329"""
    sr = 44100
    j = np.arange(sr, dtype=np.float32)
    exp = np.sin(j / np.float32(sr) * F32PI).astype(np.float32)
    o = _play(be, _sine(sr)[None, :], float(sr), length=sr,
              setup=lambda s, b: (s.playback_rate.set_value(0.5), s.start()))
    assert np.max(np.abs(o[0] - exp)) <= 1e-6
    o = _play(be, _sine(sr)[None, :], float(sr), length=sr, setup=lambda s, b: (s.detune.set_value(-1200.0), s.start()))
    assert np.max(np.abs(o[0] - exp)) <= 1e-6


def test_source_negative_playback_rate(be):
    """src/node/audio_buffer_source.rs:1258-1291"""
    sr = 44100
    sine = _sine(sr)
    o = _play(be, sine[None, :], float(sr), length=sr,
              setup=lambda s, b: (s.playback_rate.set_value(-1.0), s.start_at_with_offset(0.0, b.duration)))
    exp = sine[::-1].copy()
    exp = np.concatenate([[np.float32(0.0)], exp[:-1]])
    assert np.max(np.abs(o[0] - exp)) <= 1e-6


def test_source_end_of_file(be):
    """src/node/audio_buffer_source.rs:1332-1381, :1837-1889"""
    sr = 48000.0
    d = np.zeros((1, 129), np.float32)
    d[0, 0] = d[0, 128] = 1.0
    o = _play(be, d, sr, length=256, setup=lambda s, b: s.start_at(0.0))
    e = np.zeros(256, np.float32)
    e[0] = e[128] = 1.0
    assert np.array_equal(o[0], e)
    o = _play(be, d, sr, length=256, setup=lambda s, b: s.start_at(1.0 / sr))
    e = np.zeros(256, np.float32)
    e[1] = e[129] = 1.0
    assert np.max(np.abs(o[0] - e)) <= 1e-10
    d5 = np.zeros((1, 5), np.float32)
    d5[0, 0] = 1.0
    o = _play(be, d5, sr, setup=lambda s, b: (s.start_at(0.0), s.stop_at(125.0 / sr)))
    e = np.zeros(128, np.float32)
    e[0] = 1.0
    assert np.array_equal(o[0], e)
    o = _play(be, d5, sr, setup=lambda s, b: (s.start_at(1.0 / sr), s.stop_at(125.0 / sr)))
    e = np.zeros(128, np.float32)
    e[1] = 1.0
    assert np.array_equal(o[0], e)


def test_source_duration_and_offset(be):
    """src/node/audio_buffer_source.rs:1384-1506, :1537-1573"""
    sr = 48000.0
    d = np.zeros((1, RQ), np.float32)
    d[0, 4] = d[0, 5] = 1.0
    o = _play(be, d, sr, setup=lambda s, b: s.start_at_with_offset_and_duration(0.0, 0.0, 4.5 / sr))
    e = np.zeros(RQ, np.float32)
    e[4] = 1.0
    assert np.array_equal(o[0], e)
    o = _play(be, d, sr, setup=lambda s, b: s.start_at_with_offset_and_duration(1.0 / sr, 0.0, 4.5 / sr))
    e = np.zeros(RQ, np.float32)
    e[5] = 1.0
    assert np.array_equal(o[0], e)
    o = _play(be, d, sr, setup=lambda s, b: s.start_at_with_offset_and_duration(0.0, 1.0 / sr, 3.5 / sr))
    e = np.zeros(RQ, np.float32)
    e[3] = 1.0
    assert np.array_equal(o[0], e)
    # wpt sub-sample-grain
    sr2 = 32768.0
    start_i, end_i = 3.1, 37.2
    o = _play(be, np.ones((1, RQ), np.float32), sr2,
              setup=lambda s, b: s.start_at_with_offset_and_duration(start_i / sr2, 0.0, (end_i - start_i) / sr2))
    e = np.ones(RQ, np.float32)
    e[:int(math.floor(start_i)) + 1] = 0.0
    e[int(math.ceil(end_i)):] = 0.0
    assert np.array_equal(o[0], e)
    # reverse playback with duration
    o = _play(be, [[1.0, 2.0, 3.0, 4.0, 5.0]], sr,
              setup=lambda s, b: (s.playback_rate.set_value(-1.0),
                                  s.start_at_with_offset_and_duration(0.0, b.duration, 2.0 / sr)))
    e = np.zeros(RQ, np.float32)
    e[1] = 5.0
    assert np.array_equal(o[0], e)
    # offset larger than buffer duration (source not connected in the reference test: output silent)
    o = _play(be, np.ones((1, 13), np.float32), sr, setup=lambda s, b: s.start_at_with_offset(0.0, 64.0 / sr))
    assert np.array_equal(o[0], np.zeros(RQ, np.float32))


LOOP_LENS = [RQ // 2 - 1, RQ // 2, RQ // 2 + 1, RQ - 1, RQ, RQ + 1, RQ * 2 - 1, RQ * 2, RQ * 2 + 1]


@pytest.mark.parametrize("blen", LOOP_LENS)
def test_source_loops(be, blen):
    """src/node/audio_buffer_source.rs:1576-1756 — fast/slow track loops, mono/stereo."""
    sr, ln = 48000.0, RQ * 4
    d = np.zeros((1, blen), np.float32)
    d[0, 0] = 1.0
    o = _play(be, d, sr, length=ln, setup=lambda s, b: (s.set_loop(True), s.start()))
    e = np.zeros(ln, np.float32)
    e[0:ln:blen] = 1.0
    assert np.max(np.abs(o[0] - e)) <= 1e-10
    o = _play(be, d, sr, length=ln, setup=lambda s, b: (s.set_loop(True), s.start_at(1.0 / sr)))
    e = np.zeros(ln, np.float32)
    e[1:ln:blen] = 1.0
    assert np.max(np.abs(o[0] - e)) <= 1e-9
    d2 = np.zeros((2, blen), np.float32)
    d2[0, 0] = 1.0
    d2[1, 1] = 1.0
    for start, first, tol in [(0.0, 0, 1e-10), (1.0 / sr, 1, 1e-9)]:
        o = _play(be, d2, sr, length=ln, channels=2, setup=lambda s, b: (s.set_loop(True), s.start_at(start)))
        el, er = np.zeros(ln, np.float32), np.zeros(ln, np.float32)
        for i in range(first, ln, blen):
            el[i] = 1.0
            if i < ln - 1:
                er[i + 1] = 1.0
        assert np.max(np.abs(o[0] - el)) <= tol and np.max(np.abs(o[1] - er)) <= tol


def test_source_reverse_loop_boundaries(be):
    """src/node/audio_buffer_source.rs:1758-1778"""
    sr = 48000.0
    o = _play(be, [[1.0, 2.0, 3.0, 4.0, 5.0]], sr,
              setup=lambda s, b: (s.set_loop(True), s.set_loop_start(1.0 / sr), s.set_loop_end(4.0 / sr),
                                  s.playback_rate.set_value(-1.0), s.start_at_with_offset(0.0, 3.0 / sr)))
    assert np.array_equal(o[0, :8], np.float32([4, 3, 2, 4, 3, 2, 4, 3]))


@pytest.mark.parametrize("ls,le,err", [(-2.0, -1.0, 0.0), (-1.0, -2.0, 0.0), (0.0, 0.0, 0.0), (-1.0, 2.0, 0.0),
                                       (2.0, -1.0, 1e-10), (1.0, 1.0, 1e-10), (2.0, 3.0, 1e-10), (3.0, 2.0, 1e-10)])
def test_source_loop_out_of_bounds(be, ls, le, err):
    """src/node/audio_buffer_source.rs:1780-1835"""
    sr = 48000.0
    length = 4800
    d = np.zeros((1, 500), np.float32)
    d[0, 0] = 1.0
    o = _play(be, d, sr, length=length,
              setup=lambda s, b: (s.set_loop(True), s.set_loop_start(ls), s.set_loop_end(le), s.start()))
    e = np.zeros(length, np.float32)
    e[0:length:500] = 1.0
    assert np.max(np.abs(o[0] - e)) <= err


def test_source_start_twice(be):
    """src/node/scheduled_source.rs / audio_buffer_source.rs:300 InvalidStateError"""
    c = ctx(be, 1, RQ, 48000.0)
    s = c.create_buffer_source()
    s.start()
    with pytest.raises(waa.WaaError, match="InvalidStateError"):
        s.start()


# ----------------------------------------------------------------------------- waveshaper
def test_waveshaper(be):
    """src/node/waveshaper.rs:673-741 (tolerance 0)"""
    sr = 44100.0
    c = ctx(be, 1, 3 * RQ, sr)
    sh = c.create_wave_shaper()
    sh.set_curve([-0.5, 0.0, 0.5])
    sh.connect(c.destination())
    data = np.concatenate([np.full(RQ, -1.0), np.zeros(RQ), np.ones(RQ)]).astype(np.float32)
    s = c.create_buffer_source()
    s.connect(sh)
    s.set_buffer(buf(data[None, :], sr))
    s.start_at(0.0)
    o = c.start_rendering_sync().data[0, 0]
    assert np.array_equal(o, (data * np.float32(0.5)).astype(np.float32))

    c = ctx(be, 1, RQ, sr)
    sh = c.create_wave_shaper()
    sh.set_curve([-0.5, 0.0, 0.5])
    sh.connect(c.destination())
    x = (np.arange(RQ, dtype=np.float32) / np.float32(RQ) * np.float32(2.0) - np.float32(1.0)).astype(np.float32)
    data = np.zeros(3 * RQ, np.float32)
    data[:RQ] = x
    s = c.create_buffer_source()
    s.connect(sh)
    s.set_buffer(buf(data[None, :], sr))
    s.start_at(0.0)
    o = c.start_rendering_sync().data[0, 0]
    assert np.array_equal(o, (x / np.float32(2.0)).astype(np.float32))


# ----------------------------------------------------------------------------- convolver
def _convolve(be, signal, ir, length, normalize=True, channels=1, sr=44100.0):
    c = ctx(be, channels, length, sr)
    s = c.create_buffer_source()
    s.set_buffer(buf(signal, sr))
    s.start()
    cv = c.create_convolver(disable_normalization=not normalize)
    if ir is not None:
        cv.set_buffer(buf(ir, sr))
    s.connect(cv).connect(c.destination())
    return c.start_rendering_sync().data[0]


def test_convolver_basic(be):
    """src/node/convolver.rs:551-650"""
    cal = np.float32(0.00125)
    sig = [[0.0, 1.0, 0.0, -1.0, 0.0]]
    o = _convolve(be, sig, None, 10)
    assert np.max(np.abs(o[0] - np.float32([0, 1, 0, -1, 0, 0, 0, 0, 0, 0]))) <= 1e-6
    o = _convolve(be, sig, np.zeros((1, 0), np.float32), 10)
    assert np.max(np.abs(o[0])) <= 1e-6
    o = _convolve(be, sig, [[0.0] * 6], 10)
    assert np.max(np.abs(o[0])) <= 1e-6
    o = _convolve(be, sig, [[1.0]], 10)
    assert np.max(np.abs(o[0] - np.float32([0, cal, 0, -cal, 0, 0, 0, 0, 0, 0]))) <= 1e-6
    o = _convolve(be, sig, [[1.0, 1.0]], 10)
    assert np.max(np.abs(o[0] - np.float32([0, cal, cal, -cal, -cal, 0, 0, 0, 0, 0]))) <= 1e-6


def test_convolver_tail_time(be):
    """src/node/convolver.rs:653-668"""
    o = _convolve(be, [[1.0]], np.ones((1, 256), np.float32), 512)[0]
    assert not np.any(o[:256] <= 1e-6)
    assert np.max(np.abs(o[256:])) <= 1e-6


def test_convolver_errors(be):
    """src/node/convolver.rs:520-549"""
    c = ctx(be, 1, 128, 44100.0)
    with pytest.raises(waa.WaaError, match="NotSupportedError"):
        c.create_convolver(buffer=buf([[1.0]], 48000.0))
    c = ctx(be, 1, 128, 48000.0)
    with pytest.raises(waa.WaaError, match="NotSupportedError"):
        c.create_convolver(buffer=buf(np.ones((3, 1)), 48000.0))


@pytest.mark.parametrize("inp,ir,n_out,expect", [
    ([[1.0]], [[0.0, 1.0]], 1, {0: [1]}),
    ([[1.0]], [[0.0, 1.0, 0.0], [0.0, 0.0, 1.0]], 2, {0: [1], 1: [2]}),
    ([[1.0, 0.0], [0.0, 1.0]], [[0.0, 1.0]], 2, {0: [1], 1: [2]}),
    ([[1.0, 0.0], [0.0, 1.0]], [[0.0, 1.0, 0.0], [0.0, 0.0, 1.0]], 2, {0: [1], 1: [3]}),
    ([[1.0, 0.0], [0.0, 1.0]], [[0, 1, 0, 0, 0], [0, 0, 1, 0, 0], [0, 0, 0, 1, 0], [0, 0, 0, 0, 1]], 2,
     {0: [1, 4], 1: [2, 5]}),
    ([[1.0, 0.0]], [[0, 1, 0, 0, 0], [0, 0, 1, 0, 0], [0, 0, 0, 1, 0], [0, 0, 0, 0, 1]], 2, {0: [1, 3], 1: [2, 4]}),
])
def test_convolver_channel_routing(be, inp, ir, n_out, expect):
    """src/node/convolver.rs:671-991 — six (input, IR) channel configurations, 1e-7."""
    o = _convolve(be, inp, ir, 128, normalize=False, channels=n_out)
    for c in range(n_out):
        e = np.zeros(128, np.float32)
        for i in expect[c]:
            e[i] = 1.0
        assert np.max(np.abs(o[c] - e)) <= 1e-7


def test_fftconvolver_matches_exact_multi_partition(orc_lib):
    """Multi-partition behaviour is unpinned by the reference's tests: anchor the
    restated fft-convolver algorithm on the exact f64 linear convolution."""
    import ctypes as C
    rng = np.random.default_rng(7)
    nx, nh = 128 * 60, 3000  # 3 partitions of 1024
    x = rng.uniform(-1, 1, nx).astype(np.float32)
    h = (rng.uniform(-1, 1, nh) * np.exp(-np.arange(nh) / 700.0)).astype(np.float32) * np.float32(0.05)
    y = np.zeros(nx, np.float32)
    FP, DP = C.POINTER(C.c_float), C.POINTER(C.c_double)
    orc_lib.orc_fftconvolver_run.argtypes = [FP, C.c_uint64, FP, C.c_uint64, FP]
    orc_lib.orc_fftconvolver_run(h.ctypes.data_as(FP), nh, x.ctypes.data_as(FP), nx, y.ctypes.data_as(FP))
    ye = np.zeros(nx, np.float64)
    orc_lib.orc_convolve_exact.argtypes = [FP, C.c_uint64, FP, C.c_uint64, DP, C.c_uint64]
    orc_lib.orc_convolve_exact(x.ctypes.data_as(FP), nx, h.ctypes.data_as(FP), nh, ye.ctypes.data_as(DP), nx)
    ref = np.convolve(x.astype(np.float64), h.astype(np.float64))[:nx]
    assert np.max(np.abs(ye - ref)) <= 1e-12
    rms = np.sqrt(np.mean((y - ye) ** 2))
    assert rms <= 1e-6, rms


# ----------------------------------------------------------------------------- resample
def test_buffer_resample(be):
    """src/buffer.rs:736-817"""
    o = waa.resample(be, [[1, 2, 3, 4, 5]], 48000.0, 96000.0)
    exp = np.float32(1.0) + np.float32(4.0 / 9.0) * np.arange(10, dtype=np.float32)
    assert o.shape == (1, 10) and np.max(np.abs(o[0] - exp)) <= 1e-6
    o = waa.resample(be, [[1, 2, 3, 4, 5]], 96000.0, 48000.0)
    assert np.array_equal(o[0], np.float32([1, 3, 5]))
    for sr in (22500, 38000, 48000, 96000):
        i = np.arange(sr, dtype=np.float32)
        ph = i / np.float32(sr) * np.float32(2.0) * F32PI
        o = waa.resample(be, np.stack([np.sin(ph), np.cos(ph)]).astype(np.float32), float(sr), 44100.0)
        j = np.arange(44100, dtype=np.float32) / np.float32(44100) * np.float32(2.0) * F32PI
        assert o.shape == (2, 44100)
        assert np.max(np.abs(o[0] - np.sin(j))) <= 1e-3 and np.max(np.abs(o[1] - np.cos(j))) <= 1e-3



def test_parking_garage_ir_fixture(hip, orc):
    """BASELINE.json config 3's impulse response: samples/parking-garage-response.wav (2 ch, 44.1 kHz, 16-bit,
    164 363 frames) -> AudioBuffer::resample to 48 kHz (decoding.rs:51, buffer.rs:311-363) = 2 x 178 899 frames,
    i.e. 175 partitions of 1024.  The product's host resampler and the oracle's agree bit for bit
    (waa_buffer_resample is host code: no device needed), first and last samples are kept (buffer.rs:330-347)."""
    from graphs import garage_ir
    a, b = garage_ir(hip), garage_ir(orc)
    raw = garage_ir(hip, sr=44100.0)
    assert raw.shape == (2, 164363) and a.shape == b.shape == (2, 178899)
    assert np.array_equal(a, b)
    assert np.array_equal(a[:, 0], raw[:, 0]) and np.array_equal(a[:, -1], raw[:, -1])
    assert (a.shape[1] + 1023) // 1024 == 175
    assert float(np.abs(raw).max()) == 1.0  # a full-scale (-32768) sample: i16 / 32768

# ----------------------------------------------------------------------------- analyser
def test_blackman(orc_lib):
    """src/analysis.rs:415-437"""
    import ctypes as C
    v = np.zeros(2048, np.float32)
    orc_lib.orc_blackman.argtypes = [C.c_uint32, C.POINTER(C.c_float)]
    orc_lib.orc_blackman(2048, v.ctypes.data_as(C.POINTER(C.c_float)))
    assert 0.0 < v.min() < 0.01 and 0.99 < v.max() <= 1.0
    assert int(np.argmin(v)) == 0 and int(np.argmax(v)) == 1024


def test_analyser_peak_bin_and_silence(be):
    """src/analysis.rs:721-796 driven through a render: source -> analyser -> destination."""
    sr, fft = 44100.0, 1024
    for num_bin in (1, 7, 40, 127):
        freq = np.float32(43.066) * np.float32(num_bin)
        i = np.arange(fft, dtype=np.float32)
        sig = np.sin(freq * i / np.float32(sr) * np.float32(2.0) * F32PI).astype(np.float32)
        c = ctx(be, 1, fft, sr)
        s = c.create_buffer_source()
        s.set_buffer(buf(sig[None, :], sr))
        s.start()
        a = c.create_analyser(fft_size=fft)
        s.connect(a).connect(c.destination())
        out = c.start_rendering_sync().data[0, 0]
        assert np.array_equal(out, sig)  # passthrough
        bins = a.get_float_frequency_data()
        assert bins.shape == (fft // 2,) and int(np.argmax(bins)) == num_bin
        td = a.get_float_time_domain_data()
        assert np.array_equal(td, sig)
    c = ctx(be, 1, RQ, sr)
    a = c.create_analyser(fft_size=RQ)
    a.connect(c.destination())
    c.start_rendering_sync()
    bins = a.get_float_frequency_data(n=RQ)
    assert np.all(np.isneginf(bins[:RQ // 2])) and np.all(bins[RQ // 2:] == 0.0)
    assert np.all(a.get_byte_frequency_data() == 0)


def test_analyser_matches_numpy_dft(be):
    """dB values are 'parity unpinned' by the reference; pin the oracle to the DFT definition."""
    sr, fft, n = 48000.0, 2048, 128 * 40
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, (2, n)).astype(np.float32)
    c = ctx(be, 2, n, sr)
    s = c.create_buffer_source()
    s.set_buffer(buf(x, sr))
    s.start()
    a = c.create_analyser(fft_size=fft, smoothing_time_constant=0.8)
    s.connect(a).connect(c.destination())
    c.start_rendering_sync()
    db = a.get_float_frequency_data()
    mono = (np.float32(0.5) * (x[0] + x[1]))[-fft:]
    i = np.arange(fft, dtype=np.float64)
    w = 0.42 - 0.5 * np.cos(2 * np.pi * i / fft) + 0.08 * np.cos(4 * np.pi * i / fft)
    mag = np.abs(np.fft.rfft(mono.astype(np.float64) * w))[:fft // 2] / fft * (1 - 0.8)
    ref = 20 * np.log10(mag)
    assert np.max(np.abs(db - ref)) <= 1e-3


def test_oracle_threaded_render_is_identical(orc, orc_lib):
    """The oracle's multi-threaded mode (used by bench.py's cpu_baseline leg) must equal its serial mode."""
    import ctypes as C
    rng = np.random.default_rng(4)
    n, frames, sr = 6, 128 * 40, 48000.0
    noise = rng.uniform(-1, 1, (n, 2, frames)).astype(np.float32)
    ir = (rng.uniform(-1, 1, (2, 3000)) * np.exp(-np.arange(3000) / 600.0)).astype(np.float32)
    outs = []
    for threads in (1, 4):
        c = ctx(orc, 2, frames, sr, n_instances=n)
        s = c.create_buffer_source()
        s.set_buffer_batch(noise, sr)
        f = c.create_biquad_filter(type_="lowpass", frequency=200.0)
        cv = c.create_convolver(buffer=buf(ir, sr))
        an = c.create_analyser()
        s.connect(f).connect(cv).connect(an).connect(c.destination())
        s.start()
        c.prepare()
        orc_lib.orc_set_threads.argtypes = [C.c_void_p, C.c_int32]
        orc_lib.orc_set_threads(c._handle, threads)
        outs.append(c.start_rendering_sync().data)
    assert np.array_equal(outs[0], outs[1])


def test_constant_source_start_stop(orc):
    """src/node/constant_source.rs test_start_stop (:308-340): start in the 2nd block at frame 129, stop in the 3rd at
    frame 257, tolerance 0.  (Oracle pin; the HIP path renders constant sources with sub-quantum starts and stops
    in tests/test_delay.py::test_subquantum_delay_dynamic_lifetime and in the randomised graphs.)"""
    sr = 48000.0
    c = waa.OfflineAudioContext(1, 128 * 4, sr, binding=orc)
    src = c.create_constant_source()
    src.connect(c.destination())
    src.start_at(129.0 / sr)
    src.stop_at(257.0 / sr)
    out = c.start_rendering_sync().data[0, 0]
    exp = np.zeros(512, np.float32)
    exp[129:257] = 1.0
    assert np.array_equal(out, exp)


def test_gain_audioparam_value_applies_immediately(orc):
    """src/node/gain.rs test_audioparam_value_applies_immediately (:209-229): a value set before rendering is in
    effect from the first frame"""
    c = waa.OfflineAudioContext(1, 128, 48000.0, binding=orc)
    g = c.create_gain(gain=0.5)
    src = c.create_constant_source()
    src.connect(g).connect(c.destination())
    src.start()
    out = c.start_rendering_sync().data[0, 0]
    assert np.array_equal(out, np.full(128, 0.5, np.float32))


def _analyser_of_constant(orc, value, fft_size):
    c = waa.OfflineAudioContext(1, 128, 48000.0, binding=orc)
    src = c.create_constant_source(offset=value)
    an = c.create_analyser(fft_size=fft_size)
    src.connect(an).connect(c.destination())
    src.start()
    c.start_rendering_sync()
    return an


def test_analyser_time_domain_data_vs_fft_size(orc):
    """src/analysis.rs:656-691: a destination longer than fftSize only gets fftSize values (the rest is untouched),
    a shorter one is filled completely; tolerance 0"""
    an = _analyser_of_constant(orc, 1.0, 32)
    got = an.get_float_time_domain_data(n=128)
    exp = np.zeros(128, np.float32)
    exp[:32] = 1.0
    assert np.array_equal(got, exp)
    an = _analyser_of_constant(orc, 1.0, 128)
    assert np.array_equal(an.get_float_time_domain_data(n=16), np.ones(16, np.float32))


def test_analyser_byte_time_domain_data(orc):
    """src/analysis.rs:694-718: +1 -> 255, -1 -> 0"""
    assert np.array_equal(_analyser_of_constant(orc, 1.0, 128).get_byte_time_domain_data(n=128), np.full(128, 255, np.uint8))
    assert np.array_equal(_analyser_of_constant(orc, -1.0, 128).get_byte_time_domain_data(n=128), np.zeros(128, np.uint8))


def test_offline_start_stop_square_dc(orc):
    """tests/offline.rs:48-81: a square oscillator at 0 Hz is a constant 1; started at quantum 1, stopped at quantum 3;
    tolerance 0"""
    sr = 48000.0
    c = waa.OfflineAudioContext(1, 128 * 4, sr, binding=orc)
    osc = c.create_oscillator(type_="square", frequency=0.0)
    osc.connect(c.destination())
    osc.start_at(128.0 / sr)
    osc.stop_at(128.0 * 3.0 / sr)
    out = c.start_rendering_sync().data
    assert out.shape == (1, 1, 512)
    exp = np.zeros(512, np.float32)
    exp[128:384] = 1.0
    assert np.array_equal(out[0, 0], exp)


def test_offline_delayed_constant_source(orc):
    """tests/offline.rs:83-112: constant source through a 2-quantum delay, abs_all <= 1e-5"""
    sr = 48000.0
    c = waa.OfflineAudioContext(1, 128 * 4, sr, binding=orc)
    delay = c.create_delay(1.0)
    delay.delay_time.set_value(np.float32(128.0 * 2.0) / np.float32(sr))
    delay.connect(c.destination())
    src = c.create_constant_source()
    src.connect(delay)
    src.start()
    out = c.start_rendering_sync().data[0, 0]
    exp = np.zeros(512, np.float32)
    exp[256:] = 1.0
    assert np.max(np.abs(out - exp)) <= 1e-5
