"""AudioParam automation timeline (src/param.rs:796-1584): the reference's unit tests re-typed
(src/param.rs:1765-3275) against the stand-alone timeline object of BOTH libraries — the oracle (prefix orc_) and
the product's host-side evaluator (prefix waa_, plain C++: automation is control-side work, no device needed).
Vectors and tolerances are the reference's (abs_all <= 0 unless noted); f32 arithmetic like the reference."""
import ctypes as C

import numpy as np
import pytest

import web_audio_api_rs_amd as waa

SET, SET_AT, LIN, EXP, CANCEL, TARGET, HOLD, CURVE = range(8)
f32 = np.float32


class TL:
    def __init__(self, lib, prefix, default, lo, hi, a_rate=True):
        self.lib, self.p = lib, prefix
        fn = getattr(lib, prefix + "timeline_create")
        fn.restype = C.c_void_p
        fn.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int32]
        self.h = C.c_void_p(fn(default, lo, hi, int(a_rate)))
        self.ev = getattr(lib, prefix + "timeline_event")
        self.ev.restype = C.c_int32
        self.ev.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_double, C.c_double, C.POINTER(C.c_float), C.c_uint32]
        self.cp = getattr(lib, prefix + "timeline_compute")
        self.cp.restype = C.c_uint32
        self.cp.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_uint32, C.POINTER(C.c_float)]
        self.val = getattr(lib, prefix + "timeline_value")
        self.val.restype = C.c_float
        self.val.argtypes = [C.c_void_p]

    def event(self, kind, value=0.0, time=0.0, aux=0.0, curve=None):
        arr = None if curve is None else np.asarray(curve, np.float32)
        ptr = None if arr is None else arr.ctypes.data_as(C.POINTER(C.c_float))
        return self.ev(self.h, kind, value, time, aux, ptr, 0 if arr is None else arr.size)

    def ok(self, *a, **kw):
        assert self.event(*a, **kw) == 0

    def compute(self, block_time, count=10, dt=1.0):
        out = np.zeros(count, np.float32)
        n = self.cp(self.h, block_time, dt, count, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out[:n].copy()

    def value(self):
        return self.val(self.h)

    def __del__(self):
        fn = getattr(self.lib, self.p + "timeline_destroy")
        fn.argtypes = [C.c_void_p]
        fn(self.h)


@pytest.fixture(params=["orc", "waa"])
def mk(request, orc_lib, hip):
    lib, prefix = (orc_lib, "orc_") if request.param == "orc" else (hip.lib, "waa_")
    return lambda default, lo, hi, a_rate=True: TL(lib, prefix, default, lo, hi, a_rate)


def eq(a, b, tol=0.0):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    assert a.shape == b.shape, (a, b)
    assert np.max(np.abs(a - b)) <= tol, (a, b)


def target_curve(v0, v1, t0, tc, n):
    return [f32(v1) + f32(f32(v0) - f32(v1)) * f32(np.exp(-((t - t0) / tc))) for t in range(n)]


_libm = C.CDLL("libm.so.6")
_libm.powf.restype = C.c_float
_libm.powf.argtypes = [C.c_float, C.c_float]


def powf(x, y):
    """the C library's powf — what Rust's f32::powf resolves to (numpy's float32 power differs in the last bit)"""
    return f32(_libm.powf(f32(x), f32(y)))


def exp_curve(start, end, n, dur):
    return [f32(start) * powf(f32(end) / f32(start), f32(t) / f32(dur)) for t in range(n)]


def test_set_value(mk):  # :1765
    t = mk(0.0, -10.0, 10.0)
    t.ok(SET, 2.0)
    assert t.value() == 2.0
    eq(t.compute(0.0), [2.0] * 10)
    t = mk(0.0, 0.0, 1.0)
    t.ok(SET, 2.0)
    assert t.value() == 1.0          # value() is clamped, the intrinsic value is not
    eq(t.compute(0.0), [2.0] * 10)
    assert t.value() == 1.0


def test_steps_a_rate(mk):  # :1814
    t = mk(0.0, -10.0, 10.0)
    t.ok(SET_AT, 5.0, 2.0)
    t.ok(SET_AT, 12.0, 8.0)
    t.ok(SET_AT, 8.0, 10.0)
    eq(t.compute(0.0), [0, 0, 5, 5, 5, 5, 5, 5, 12, 12])
    eq(t.compute(10.0), [8.0])
    t = mk(0.0, -10.0, 10.0)
    t.ok(SET_AT, 5.0, 2.0)
    t.ok(SET_AT, 8.0, 12.0)
    eq(t.compute(0.0), [0, 0, 5, 5, 5, 5, 5, 5, 5, 5])
    eq(t.compute(10.0), [5, 5, 8, 8, 8, 8, 8, 8, 8, 8])


def test_steps_k_rate(mk):  # :1874
    t = mk(0.0, -10.0, 10.0, a_rate=False)
    for v, tm in ((5.0, 2.0), (12.0, 8.0), (8.0, 10.0), (3.0, 14.0)):
        t.ok(SET_AT, v, tm)
    eq(t.compute(0.0), [0.0])
    eq(t.compute(10.0), [8.0])
    eq(t.compute(20.0), [3.0])


def test_linear_ramp_a_rate(mk):  # :1901-1958
    t = mk(0.0, -10.0, 10.0)
    t.ok(SET_AT, 5.0, 2.0)
    t.ok(LIN, 8.0, 5.0)
    t.ok(LIN, 0.0, 13.0)
    eq(t.compute(0.0), [0, 0, 5, 6, 7, 8, 7, 6, 5, 4])
    t = mk(0.0, -10.0, 10.0)
    t.ok(SET_AT, 0.0, 0.0)
    t.ok(LIN, 9.0, 9.0)
    eq(t.compute(0.0), list(range(10)))


def test_linear_ramp_implicit_set_value_and_multiple_blocks(mk):  # :1959-2034
    t = mk(0.0, -10.0, 10.0)
    eq(t.compute(0.0), [0.0])
    t.ok(LIN, 10.0, 20.0)
    eq(t.compute(10.0), list(range(10)))
    eq(t.compute(20.0), [10.0] * 10)
    t = mk(0.0, -20.0, 20.0)
    t.ok(LIN, 20.0, 20.0)
    eq(t.compute(0.0), list(range(10)))
    assert t.value() == 0.0
    eq(t.compute(10.0), list(range(10, 20)))
    assert t.value() == 10.0
    eq(t.compute(20.0), [20.0] * 10)
    assert t.value() == 20.0


def test_linear_ramp_k_rate_multiple_blocks(mk):  # :2035-2093
    t = mk(0.0, -20.0, 20.0, a_rate=False)
    t.ok(LIN, 20.0, 20.0)
    for bt, v in ((0.0, 0.0), (10.0, 10.0), (20.0, 20.0)):
        eq(t.compute(bt), [v])
        assert t.value() == v
    t = mk(0.0, -20.0, 20.0, a_rate=False)
    t.ok(LIN, 15.0, 15.0)
    for bt, v in ((0.0, 0.0), (10.0, 10.0), (20.0, 15.0)):
        eq(t.compute(bt), [v])


def test_linear_ramp_start_time(mk):  # :2094
    t = mk(0.0, -10.0, 10.0)
    t.ok(SET_AT, 1.0, 0.0)
    t.ok(LIN, -1.0, 10.0)
    eq(t.compute(0.0), [1, 0.8, 0.6, 0.4, 0.2, 0, -0.2, -0.4, -0.6, -0.8], 1e-7)
    eq(t.compute(10.0), [-1.0] * 10)
    t.ok(LIN, 1.0, 30.0)  # starts at the end of the last event (10.)
    eq(t.compute(20.0), [0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9], 1e-7)


def test_exponential_ramp_a_rate(mk):  # :2130-2204
    t = mk(0.0, 0.0, 1.0)
    t.ok(SET_AT, 0.0001, 0.0)
    t.ok(EXP, 1.0, 10.0)
    eq(t.compute(0.0), exp_curve(0.0001, 1.0, 10, 10))
    eq(t.compute(10.0), [1.0] * 10)
    t = mk(0.0, 0.0, 1.0)
    t.ok(SET_AT, 0.0001, 3.0)
    t.ok(EXP, 1.0, 13.0)
    res = [0.0] * 3 + exp_curve(0.0001, 1.0, 10, 10) + [1.0] * 7
    eq(t.compute(0.0), res[:10])
    assert t.value() == res[0]
    eq(t.compute(10.0), res[10:20])
    assert t.value() == res[10]


def test_exponential_ramp_zero_and_opposite_target(mk):  # :2205-2258, :2315-2361
    t = mk(0.0, 0.0, 1.0)
    t.ok(SET_AT, 0.0, 0.0)
    t.ok(EXP, 1.0, 5.0)
    eq(t.compute(0.0), [0, 0, 0, 0, 0, 1, 1, 1, 1, 1])
    t = mk(0.0, -1.0, 1.0)
    t.ok(SET_AT, -1.0, 0.0)
    t.ok(EXP, 1.0, 5.0)
    eq(t.compute(0.0), [-1, -1, -1, -1, -1, 1, 1, 1, 1, 1])
    t = mk(0.0, 0.0, 1.0, a_rate=False)
    t.ok(EXP, 1.0, 5.0)
    eq(t.compute(0.0), [0.0])
    eq(t.compute(10.0), [1.0])
    t = mk(-1.0, -1.0, 1.0, a_rate=False)
    t.ok(EXP, 1.0, 5.0)
    eq(t.compute(0.0), [-1.0])
    eq(t.compute(10.0), [1.0])


def test_exponential_ramp_to_zero_is_refused(mk):  # :2259
    t = mk(1.0, 0.0, 1.0)
    assert t.event(EXP, 0.0, 10.0) == 1


def test_exponential_ramp_k_rate_and_start_time(mk):  # :2274-2313, :2362-2400
    t = mk(0.0, 0.0, 1.0, a_rate=False)
    t.ok(SET_AT, 0.0001, 3.0)
    t.ok(EXP, 1.0, 13.0)
    res = [0.0] * 3 + exp_curve(0.0001, 1.0, 10, 10) + [1.0] * 7
    eq(t.compute(0.0), [res[0]])
    eq(t.compute(10.0), [res[10]])
    eq(t.compute(20.0), [1.0])
    t = mk(0.0, -10.0, 10.0)
    t.ok(SET_AT, 0.0, 0.0)
    t.ok(LIN, 1.0, 10.0)
    eq(t.compute(0.0), [0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9], 1e-7)
    eq(t.compute(10.0), [1.0] * 10)
    t.ok(EXP, 0.0001, 30.0)
    eq(t.compute(20.0), exp_curve(1.0, 0.0001, 20, 20)[10:], 1e-7)


def test_set_target_a_rate(mk):  # :2402-2512
    t = mk(0.0, 0.0, 1.0)
    t.ok(SET_AT, 0.0, 0.0)
    t.ok(TARGET, 1.0, 0.0, 1.0)
    eq(t.compute(0.0), target_curve(0.0, 1.0, 0.0, 1.0, 10))
    t = mk(0.0, 0.0, 1.0)  # implicit SetValue when SetTarget comes first
    t.ok(TARGET, 1.0, 0.0, 1.0)
    eq(t.compute(0.0), target_curve(0.0, 1.0, 0.0, 1.0, 10))
    t = mk(0.0, 0.0, 100.0)
    t.ok(SET_AT, 1.0, 1.0)
    t.ok(TARGET, 42.0, 1.0, 2.1)
    res = target_curve(1.0, 42.0, 1.0, 2.1, 10)
    res[0] = 0.0
    eq(t.compute(0.0), res)
    t = mk(0.0, 0.0, 100.0)  # time_constant == 0: jumps
    t.ok(TARGET, 1.0, 1.0, 0.0)
    eq(t.compute(0.0), [0.0] + [1.0] * 9)


def test_set_target_multiple_blocks_and_followed_by_set_value(mk):  # :2513-2588
    t = mk(0.0, 0.0, 2.0)
    t.ok(SET_AT, 0.0, 0.0)
    t.ok(TARGET, 2.0, 0.0, 1.0)
    res = target_curve(0.0, 2.0, 0.0, 1.0, 20)
    eq(t.compute(0.0), res[:10])
    eq(t.compute(10.0), res[10:])
    t = mk(0.0, 0.0, 2.0)
    t.ok(SET_AT, 0.0, 0.0)
    t.ok(TARGET, 2.0, 0.0, 1.0)
    t.ok(SET_AT, 0.5, 15.0)
    res = target_curve(0.0, 2.0, 0.0, 1.0, 15) + [0.5] * 5
    eq(t.compute(0.0), res[:10])
    eq(t.compute(10.0), res[10:])


def test_set_target_ends_at_threshold_and_waits_for_start(mk):  # :2589-2644
    t = mk(0.0, 0.0, 2.0)
    t.ok(SET_AT, 1.0, 0.0)
    t.ok(TARGET, 0.0, 1.0, 0.2)
    vs = t.compute(0.0, count=128)
    assert not np.any((vs != 0) & (np.abs(vs) < np.finfo(np.float32).tiny))  # no subnormals
    eq(t.compute(10.0, count=128), [0.0] * 128)
    t = mk(0.0, 0.0, 2.0)
    t.ok(SET_AT, 1.0, 0.0)
    t.ok(TARGET, 0.0, 5.0, 1.0)
    eq(t.compute(0.0)[:6], [1.0] * 6)


def test_set_target_followed_by_ramp(mk):  # :2645-2698
    t = mk(0.0, 0.0, 10.0)
    t.ok(SET_AT, 0.0, 0.0)
    t.ok(TARGET, 2.0, 0.0, 10.0)
    res = target_curve(0.0, 2.0, 0.0, 10.0, 11)
    eq(t.compute(0.0), res[:10])
    v0 = res[10]
    t.ok(LIN, 10.0, 20.0)
    ramp = [f32(v0) + f32(f32(10.0) - f32(v0)) * f32(tt - 10.0) / f32(10.0) for tt in range(10, 20)]
    eq(t.compute(10.0), ramp, 1e-6)
    eq(t.compute(20.0), [10.0] * 10)


def test_set_target_k_rate_and_snap_to_value(mk):  # :2699-2776
    t = mk(0.0, 0.0, 2.0, a_rate=False)
    t.ok(SET_AT, 0.0, 0.0)
    t.ok(TARGET, 2.0, 0.0, 1.0)
    res = target_curve(0.0, 2.0, 0.0, 1.0, 20)
    eq(t.compute(0.0), [res[0]])
    eq(t.compute(10.0), [res[10]])
    t = mk(0.0, 0.0, 1.0)
    t.ok(SET_AT, 1.0, 0.0)
    t.ok(TARGET, 0.0, 0.0, 1.0)
    res = target_curve(1.0, 0.0, 0.0, 1.0, 30)
    eq(t.compute(0.0), res[:10])
    eq(t.compute(10.0), res[10:20])
    eq(t.compute(20.0), res[20:30])
    eq(t.compute(30.0), [0.0] * 10)  # snapped (|target - value| < 1e-10)


def test_cancel_scheduled_values(mk):  # :2777-2903
    t = mk(0.0, 0.0, 10.0)
    for k in range(10):
        t.ok(SET_AT, float(k), float(k))
    t.ok(CANCEL, 0.0, 5.0)
    eq(t.compute(0.0), [0, 1, 2, 3, 4, 4, 4, 4, 4, 4])
    t = mk(0.0, 0.0, 10.0)
    t.ok(SET_AT, 0.0, 0.0)
    t.ok(LIN, 10.0, 10.0)
    t.ok(CANCEL, 0.0, 10.0)
    eq(t.compute(0.0), [0.0] * 10)
    t = mk(0.0, 0.0, 20.0)  # ramp already started: back to the previous value
    t.ok(SET_AT, 0.0, 0.0)
    t.ok(LIN, 20.0, 20.0)
    eq(t.compute(0.0), list(range(10)))
    t.ok(CANCEL, 0.0, 10.0)
    eq(t.compute(10.0), [0.0])
    t = mk(0.0, 0.0, 10.0)
    t.ok(LIN, 10.0, 10.0)
    t.ok(CANCEL, 0.0, 10.0)
    eq(t.compute(0.0), [0.0] * 10)
    t = mk(0.0, 0.0, 20.0)
    t.ok(LIN, 20.0, 20.0)
    eq(t.compute(0.0), list(range(10)))
    t.ok(CANCEL, 0.0, 10.0)
    eq(t.compute(10.0), [0.0])


def test_cancel_and_hold(mk):  # :2904-3143
    t = mk(0.0, 0.0, 10.0)
    for k in (1, 2, 3, 4):
        t.ok(SET_AT, float(k), float(k))
    t.ok(HOLD, 0.0, 2.5)
    eq(t.compute(0.0), [0, 1, 2, 2, 2, 2, 2, 2, 2, 2])
    t = mk(0.0, 0.0, 2.0)  # during a SetTarget
    t.ok(SET_AT, 0.0, 0.0)
    t.ok(TARGET, 2.0, 0.0, 1.0)
    t.ok(HOLD, 0.0, 15.0)
    res = target_curve(0.0, 2.0, 0.0, 1.0, 16)
    hold = res.pop()
    res += [hold] * 5
    eq(t.compute(0.0), res[:10])
    eq(t.compute(10.0), res[10:])
    t = mk(0.0, 0.0, 10.0)  # during a linear ramp
    t.ok(LIN, 10.0, 10.0)
    t.ok(HOLD, 0.0, 5.0)
    eq(t.compute(0.0), [0, 1, 2, 3, 4, 5, 5, 5, 5, 5])
    t = mk(0.0, 0.0, 10.0)
    t.ok(LIN, 10.0, 10.0)
    t.ok(HOLD, 0.0, 4.5)
    eq(t.compute(0.0), [0, 1, 2, 3, 4, 4.5, 4.5, 4.5, 4.5, 4.5])
    t = mk(0.0, 0.0, 10.0)  # during an exponential ramp
    t.ok(SET_AT, 0.0001, 0.0)
    t.ok(EXP, 1.0, 10.0)
    t.ok(HOLD, 0.0, 5.0)
    res = exp_curve(0.0001, 1.0, 6, 10)
    eq(t.compute(0.0), res[:5] + [res[5]] * 5)
    t = mk(0.0, 0.0, 10.0)
    t.ok(SET_AT, 0.0001, 0.0)
    t.ok(EXP, 1.0, 10.0)
    t.ok(HOLD, 0.0, 4.5)
    hold = f32(0.0001) * powf(f32(1.0) / f32(0.0001), f32(4.5) / f32(10.0))
    eq(t.compute(0.0), exp_curve(0.0001, 1.0, 5, 10) + [hold] * 5)
    curve = [0.0, 0.5, 1.0, 0.5, 0.0]  # during a value curve
    t = mk(0.0, 0.0, 2.0)
    t.ok(CURVE, 0.0, 0.0, 10.0, curve)
    t.ok(HOLD, 0.0, 5.0)
    eq(t.compute(0.0), [0, 0.2, 0.4, 0.6, 0.8, 1, 1, 1, 1, 1], 1e-7)
    t = mk(0.0, 0.0, 2.0)
    t.ok(CURVE, 0.0, 0.0, 10.0, curve)
    t.ok(HOLD, 0.0, 4.5)
    eq(t.compute(0.0), [0, 0.2, 0.4, 0.6, 0.8, 0.9, 0.9, 0.9, 0.9, 0.9], 1e-7)


def test_set_value_curve(mk):  # :3144-3276
    curve = [0.0, 0.5, 1.0, 0.5, 0.0]
    t = mk(0.0, 0.0, 10.0)
    t.ok(CURVE, 0.0, 0.0, 10.0, curve)
    eq(t.compute(0.0), [0, 0.2, 0.4, 0.6, 0.8, 1, 0.8, 0.6, 0.4, 0.2], 1e-7)
    eq(t.compute(10.0), [0.0] * 10)
    t = mk(0.0, 0.0, 10.0)
    t.ok(CURVE, 0.0, 0.0, 20.0, curve)
    eq(t.compute(0.0), [0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9], 1e-7)
    eq(t.compute(10.0), [1, 0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3, 0.2, 0.1], 1e-7)
    eq(t.compute(20.0), [0.0] * 10)
    t = mk(1.0, 0.0, 1.0)  # a curve over another event / an event inside a curve: NotSupportedError
    t.ok(SET_AT, 0.0, 5.0)
    assert t.event(CURVE, 0.0, 0.0, 10.0, curve) == 2
    t = mk(1.0, 0.0, 1.0)
    t.ok(CURVE, 0.0, 0.0, 10.0, curve)
    assert t.event(SET_AT, 0.0, 5.0) == 2
    t = mk(0.0, 0.0, 10.0)
    t.ok(CURVE, 0.0, 5.0, 10.0, curve)
    eq(t.compute(0.0), [0, 0, 0, 0, 0, 0, 0.2, 0.4, 0.6, 0.8])


def test_argument_validation(mk):  # :24-62, :1666-1697
    t = mk(0.0, 0.0, 1.0)
    assert t.event(CURVE, 0.0, 0.0, 1.0, [1.0]) == 3           # sequence length < 2
    assert t.event(CURVE, 0.0, 0.0, 0.0, [0.0, 1.0]) == 1       # duration must be > 0
    assert t.event(SET_AT, float("nan"), 0.0) == 1
    assert t.event(SET_AT, 1.0, -1.0) == 1
    assert t.event(LIN, 1.0, float("inf")) == 1


# --------------------------------------------------------------------------- through a render (GPU vs oracle)
from graphs import rms_err, white_noise  # noqa: E402

RQ = 128


def _automated_graph(binding, noise):
    n, _, frames = noise.shape
    sr = 48000.0
    c = waa.OfflineAudioContext(2, frames, sr, n_instances=n, binding=binding)
    src = c.create_buffer_source()
    src.set_buffer_batch(noise, sr)
    g = c.create_gain(gain=0.2)
    g.gain.set_value_at_time(0.2, 0.0).linear_ramp_to_value_at_time(1.0, 0.01).set_target_at_time(0.1, 0.015, 0.004)
    g.gain.cancel_and_hold_at_time(0.03)
    g.gain.set_value_curve_at_time([0.5, 1.0, 0.25, 0.75], 0.04, 0.02)
    bq = c.create_biquad_filter(type_="lowpass", frequency=300.0, q=2.0)
    bq.frequency.set_value_at_time(300.0, 0.0).exponential_ramp_to_value_at_time(6000.0, 0.05)
    bq.frequency.linear_ramp_to_value_at_time(800.0, 0.08)
    d = c.create_delay(0.05, delay_time=0.001)
    d.delay_time.linear_ramp_to_value_at_time(0.02, 0.06)
    pan = c.create_stereo_panner(pan=-1.0)
    pan.pan.set_target_at_time(1.0, 0.0, 0.02)
    src.connect(g).connect(bq).connect(d).connect(pan).connect(c.destination())
    src.start()
    out = c.start_rendering_sync().data
    c.close()
    return out


@pytest.mark.gpu
def test_render_with_scheduled_automation(hip, orc):
    """every event kind on Gain / Biquad / Delay / StereoPanner params: the product's host-side timeline + device
    kernels against the oracle's timeline + per-quantum render"""
    noise = white_noise(2, 2, 2048 * 2 + 500, seed0=77)
    g, o = _automated_graph(hip, noise), _automated_graph(orc, noise)
    assert rms_err(g, o).max() <= 1e-6
    assert np.abs(g - o).max() <= 5e-6


@pytest.mark.gpu
def test_set_value_after_scheduled_events_is_one_more_event(hip, orc):
    """AudioParam::set_value (param.rs:402-425) AFTER automation methods enqueues a SetValue event; through the C ABI
    that is waa_set_param_const following waa_param_schedule_event: the new constant must reach the timeline that
    already exists (it used to be dropped: the timeline kept the constant it was seeded with)."""
    noise = white_noise(2, 1, RQ * 12, seed0=5)
    outs = []
    for b in (hip, orc):
        c = waa.OfflineAudioContext(1, RQ * 12, 48000.0, n_instances=2, binding=b)
        src = c.create_buffer_source()
        src.set_buffer_batch(noise, 48000.0)
        g = c.create_gain(gain=0.25)
        g.gain.set_value_at_time(0.5, RQ * 6 / 48000.0)
        src.connect(g).connect(c.destination())
        src.start()
        c.prepare()
        b.check(b.set_param_const(c._handle, g.id, 0, 1, 0.75))  # instance 1 only, after the event above
        outs.append(c.start_rendering_sync().data)
        c.close()
    assert np.array_equal(outs[0], outs[1])
    g_ = outs[0]
    assert np.allclose(g_[0, 0, :RQ * 6], 0.25 * noise[0, 0, :RQ * 6]) and np.allclose(g_[1, 0, :RQ * 6], 0.75 * noise[1, 0, :RQ * 6])
    assert np.allclose(g_[1, 0, RQ * 6:], 0.5 * noise[1, 0, RQ * 6:])


def test_scheduling_errors_reach_the_caller(be):
    c = waa.OfflineAudioContext(1, 128, 48000.0, binding=be)
    g = c.create_gain()
    g.gain.exponential_ramp_to_value_at_time(0.0, 1.0)   # RangeError in the reference (param.rs:474)
    g.connect(c.destination())
    with pytest.raises(waa.WaaError, match="RangeError"):
        c.prepare()



def test_value_curve_sampled_before_its_start_follows_the_reference_arithmetic(mk):
    """param.rs:1470-1478 with :116-119: when a SetValueCurve is reached in a block that ends BEFORE the curve starts
    (an earlier event of the same block was consumed first), the intrinsic value kept for the next block is the
    curve sampled at next_block_time < start_time.  `position as usize` saturates to 0 in Rust and the phase is the
    fractional part of the NEGATIVE position: (3 - 1) * 0.5 + 1 = 2, not the 0.5 the SetValueAtTime left behind.
    A quirk of the reference, reproduced (both restatements used to index the curve at -3 here)."""
    p = mk(0.0, -10.0, 10.0)
    p.ok(SET_AT, 0.5, 2.0)
    p.ok(CURVE, 0.0, 20.0, 4.0, curve=[1.0, 3.0])
    eq(p.compute(0.0), [0.0, 0.0, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5])
    eq(p.compute(10.0), [2.0])  # constant block (the curve starts at next_block_time): the intrinsic value
    eq(p.compute(20.0)[:5], [1.0, 1.5, 2.0, 2.5, 3.0])


def test_ramp_without_a_consumed_event_does_not_crash(mk):
    """A ramp inserted BEFORE an already queued first event has no last event to start from; the reference panics on
    `last_event.unwrap()` (param.rs:1107).  Both restatements read a zero event instead and keep rendering."""
    p = mk(0.25, -10.0, 10.0)
    p.ok(CURVE, 0.0, 20.0, 1.0, curve=[0.5, -0.5])
    p.ok(LIN, 4.0, 8.0)  # queue not empty: no implicit SetValue in front of the ramp
    eq(p.compute(0.0), [0.0, 0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 3.5, 4.0, 4.0])


# --------------------------------------------------------------------------- differential fuzz (CPU only)
def _random_schedule(rng, tls, horizon):
    """the same random automation calls on every timeline in `tls`; returns the list of status codes"""
    statuses = []
    t = 0.0
    for _ in range(int(rng.integers(1, 9))):
        kind = int(rng.choice([SET, SET_AT, LIN, EXP, TARGET, CURVE, CANCEL, HOLD], p=[.08, .2, .2, .14, .14, .1, .07, .07]))
        t += float(rng.choice([0.0, 0.25, 1.0, 3.0, 7.5, 20.0]))
        value = float(rng.choice([-2.0, -0.5, 0.0, 1e-3, 0.5, 1.0, 3.0]))  # EXP to/from 0 and opposite signs included
        aux, curve = 0.0, None
        if kind == TARGET:
            aux = float(rng.choice([0.0, 0.5, 4.0, 30.0]))  # time constant (0: jump)
        if kind == CURVE:
            curve = rng.uniform(-1.0, 1.0, int(rng.integers(2, 7))).astype(np.float32)
            aux = float(rng.choice([1.0, 4.0, 10.5]))  # duration
        when = t if rng.random() < 0.85 else max(0.0, t - float(rng.uniform(0.0, 6.0)))  # sometimes out of order
        statuses.append(tuple(tl.event(kind, value, when, aux, curve) for tl in tls))
    return statuses


@pytest.mark.parametrize("seed", range(300))
def test_random_schedules_agree_between_the_two_restatements(orc_lib, hip, seed):
    """The oracle's timeline (plain C) and the product's (C++) are written independently from src/param.rs:796-1584;
    random event lists — ramps to and from zero, out-of-order insertion, overlapping curves (refused), cancels and
    holds in the middle of ramps — have to give the same status codes and bit-identical blocks, a-rate and k-rate,
    including the intrinsic value after every block."""
    rng = np.random.default_rng(seed)
    a_rate = bool(rng.integers(0, 2))
    lo, hi = (-1.5, 2.5) if rng.random() < 0.5 else (-3.4028235e38, 3.4028235e38)
    tls = [TL(orc_lib, "orc_", 0.25, lo, hi, a_rate), TL(hip.lib, "waa_", 0.25, lo, hi, a_rate)]
    horizon = 80
    done = 0
    while done < horizon:
        st = _random_schedule(rng, tls, horizon)
        for s in st:
            assert s[0] == s[1], (seed, st)
        n = int(rng.choice([8, 16]))
        o = tls[0].compute(float(done), count=n)
        p = tls[1].compute(float(done), count=n)
        assert o.shape == p.shape, (seed, done)
        assert np.array_equal(o, p, equal_nan=True), (seed, done, o, p)
        vo, vp = tls[0].value(), tls[1].value()
        assert vo == vp or (np.isnan(vo) and np.isnan(vp)), (seed, done)
        done += n


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(300))
def test_device_replay_of_random_schedules(hip, seed):
    """waa_timeline.hip: the kernel that replays per-instance automation on the device, against the host restatement
    (waa_timeline_compute, itself pinned by the reference's unit tests above and by the oracle's independent twin) on the
    same 300 random event lists: identical slice lengths (1 or 128) in every quantum; bit-identical values for set-value
    events, linear ramps and value curves, and within 4 ulp where powf / exp of the device math library are involved
    (exponential ramps, set-target)."""
    rng = np.random.default_rng(seed)
    a_rate = bool(rng.integers(0, 2))
    lo, hi = -3.4028235e38, 3.4028235e38
    tls = [TL(hip.lib, "waa_", 0.25, lo, hi, a_rate), TL(hip.lib, "waa_", 0.25, lo, hi, a_rate)]
    kinds = set()
    for _ in range(int(rng.integers(1, 4))):
        # (record which event kinds the script holds: the tolerance depends on it)
        state = rng.bit_generator.state
        probe = np.random.default_rng(0)
        probe.bit_generator.state = state
        for _ in range(int(probe.integers(1, 9))):
            kinds.add(int(probe.choice([SET, SET_AT, LIN, EXP, TARGET, CURVE, CANCEL, HOLD], p=[.08, .2, .2, .14, .14, .1, .07, .07])))
            probe.choice([0.0, 0.25, 1.0, 3.0, 7.5, 20.0]); probe.choice([-2.0, -0.5, 0.0, 1e-3, 0.5, 1.0, 3.0])
            break  # (only a hint; the tolerance below is decided from the values)
        _random_schedule(rng, tls, 80)
    sr, nq = 16.0, 14  # 8 s per render quantum: the schedules span the first ~10 quanta
    fn = hip.lib.waa_timeline_render_device
    fn.restype = C.c_int32
    fn.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_uint8)]
    dev = np.zeros(nq * 128, np.float32)
    lens = np.zeros(nq, np.uint8)
    assert fn(tls[0].h, nq, sr, dev.ctypes.data_as(C.POINTER(C.c_float)), lens.ctypes.data_as(C.POINTER(C.c_uint8))) == 0
    for q in range(nq):
        ref = tls[1].compute(q * 128 / sr, count=128, dt=1.0 / sr)
        assert lens[q] == (1 if ref.size == 1 else 128), (seed, q)
        d = dev[q * 128:(q + 1) * 128]
        r = np.full(128, ref[0], np.float32) if ref.size == 1 else ref
        if np.array_equal(d, r, equal_nan=True):
            continue
        # powf / exp differ by an ulp or two between the device and the host math libraries
        tol = 4 * np.spacing(np.maximum(np.abs(r), np.float32(1e-30)).astype(np.float32))
        bad = ~(np.abs(d.astype(np.float64) - r) <= tol) & ~(np.isnan(d) & np.isnan(r))
        assert not bad.any(), (seed, q, d[bad][:4], r[bad][:4])


@pytest.mark.gpu
def test_per_instance_automation_is_replayed_on_the_device(hip, orc):
    """Every context of the batch schedules its OWN automation on a-rate params (gain envelope, biquad sweep, pan, delay
    time, constant offset, oscillator glide): the planner uploads the event queues and timeline_kernel evaluates them
    (the plan says so); output against the oracle, whose timelines are evaluated per quantum on the host."""
    n, frames, sr = 4, RQ * 70 + 9, 48000.0
    noise = white_noise(n, 2, frames, seed0=91)
    outs = []
    for b in (hip, orc):
        c = waa.OfflineAudioContext(2, frames, sr, n_instances=n, binding=b)
        src = c.create_buffer_source()
        src.set_buffer_batch(noise, sr)
        bq = c.create_biquad_filter(type_="lowpass", frequency=800.0, q=2.0)
        g = c.create_gain(gain=0.2)
        pan = c.create_stereo_panner(pan=0.0)
        d = c.create_delay(0.05, delay_time=0.004)
        k = c.create_constant_source(offset=0.1)
        osc = c.create_oscillator(type_="sine", frequency=200.0)
        t_end = frames / sr
        for i in range(n):
            g.gain.set_value_at_time(0.05 + 0.1 * i, 0.0, instance=i)
            g.gain.linear_ramp_to_value_at_time(0.9 - 0.1 * i, t_end * (0.4 + 0.1 * i), instance=i)
            g.gain.set_target_at_time(0.3, t_end * 0.7, 0.01 + 0.005 * i, instance=i)
            bq.frequency.set_value_at_time(200.0 + 100.0 * i, 0.0, instance=i)
            bq.frequency.exponential_ramp_to_value_at_time(4000.0 + 500.0 * i, t_end, instance=i)
            pan.pan.set_value_curve_at_time(np.float32([-1.0, 0.5 - 0.3 * i, 1.0, 0.0]), t_end * 0.1, t_end * 0.6, instance=i)
            d.delay_time.set_value_at_time(0.002 + 0.001 * i, 0.0, instance=i)
            d.delay_time.linear_ramp_to_value_at_time(0.02, t_end * 0.9, instance=i)
            k.offset.set_value_at_time(0.1 * i, t_end * 0.2, instance=i)
            k.offset.cancel_and_hold_at_time(t_end * 0.5, instance=i)
            osc.frequency.linear_ramp_to_value_at_time(300.0 + 50.0 * i, t_end * 0.8, instance=i)
        src.connect(bq).connect(g).connect(pan).connect(c.destination())
        src.connect(d).connect(c.destination())
        k.connect(c.destination())
        osc.connect(c.destination())
        src.start()
        k.start()
        osc.start()
        if b is hip:
            plan = c.plan_describe()
            assert plan.count("replayed on the device") == 6, plan
        outs.append(c.start_rendering_sync().data)
        c.close()
    assert rms_err(*outs).max() <= 1e-6
    assert np.abs(outs[0] - outs[1]).max() <= 2e-5


def test_automated_param_advances_while_its_owner_is_idle(be):
    """An AudioParam is its own graph node and is processed in EVERY quantum (param.rs:686-699), also while the node
    that owns it sees a silent input and returns early (stereo_panner.rs:229-232).  The timeline is stateful — a
    SetValueAtTime at time 0 takes the time of the block that consumes it (param.rs:1060-1062) — so evaluating it
    only when the owner reads it shifts the whole automation by the idle quanta (the oracle used to do that; the
    randomised graphs with automation found it).  A constant 1 that starts at frame 300 into a StereoPanner whose pan
    ramps -1 -> 1 over the render: the output IS the gain pair of the pan value at absolute time t."""
    sr, frames, start = 48000.0, 128 * 40, 300
    c = waa.OfflineAudioContext(2, frames, sr, binding=be)
    k = c.create_constant_source(offset=1.0)
    pan = c.create_stereo_panner(pan=0.0)
    t_end = frames / sr
    pan.pan.set_value_at_time(-1.0, 0.0).linear_ramp_to_value_at_time(1.0, t_end)
    k.connect(pan).connect(c.destination())
    k.start_at(start / sr)
    out = c.start_rendering_sync().data[0]
    t = np.arange(frames) / sr
    x = (np.clip(-1.0 + 2.0 * t / t_end, -1.0, 1.0) + 1.0) / 2.0
    exp_l, exp_r = np.sin((1.0 - x) * np.pi / 2.0), np.sin(x * np.pi / 2.0)
    assert np.all(out[:, :start] == 0.0)
    assert np.max(np.abs(out[0, start:] - exp_l[start:])) <= 2e-6
    assert np.max(np.abs(out[1, start:] - exp_r[start:])) <= 2e-6
