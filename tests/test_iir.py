"""IIRFilterNode: the reference's own tests re-typed (src/node/iir_filter.rs:407-870),
run against the oracle (CPU suite) and the HIP path (-m gpu), plus GPU-vs-oracle parity of the streaming
IIR kernel on seeded inputs.
"""
import numpy as np
import pytest
from scipy import signal

import web_audio_api_rs_amd as waa
from graphs import rms_err, white_noise

RQ = 128
TOL = 1e-6

# src/node/iir_filter.rs:600-747: (type, feedback, feedforward) of a biquad at 2000 Hz, Q 1, gain 3 dB, 44.1 kHz
BIQUAD_EQUIVALENTS = {
    "lowpass": ([1.1252702717383296, -1.9193504546709936, 0.8747297282616704],
                [0.02016238633225159, 0.04032477266450318, 0.02016238633225159]),
    "highpass": ([1.1252702717383296, -1.9193504546709936, 0.8747297282616704],
                 [0.9798376136677485, -1.959675227335497, 0.9798376136677485]),
    "bandpass": ([1.1405555566658274, -1.9193504546709936, 0.8594444433341726],
                 [0.14055555666582747, 0.0, -0.14055555666582747]),
    "notch": ([1.1405555566658274, -1.9193504546709936, 0.8594444433341726],
              [1.0, -1.9193504546709936, 1.0]),
    "allpass": ([1.1405555566658274, -1.9193504546709936, 0.8594444433341726],
                [0.8594444433341726, -1.9193504546709936, 1.1405555566658274]),
    "peaking": ([1.1182627625098631, -1.9193504546709936, 0.8817372374901369],
                [1.167050592175986, -1.9193504546709936, 0.8329494078240139]),
    "lowshelf": ([2.8028072429836723, -4.577507200153761, 1.935999047828101],
                 [2.9011403634599007, -4.544236234748791, 1.8709368927568424]),
    "highshelf": ([2.4410054070459357, -3.8234982904056865, 1.5741972118903644],
                  [3.331142651362703, -5.440377503491735, 2.300939180659645]),
}


def ctx(be, channels, length, sr, **kw):
    return waa.OfflineAudioContext(channels, length, sr, binding=be, **kw)


# --------------------------------------------------------------------------- reference KATs (both backends)
def test_constructor_validation(be):
    """iir_filter.rs:441-482: 21 coefficients / all-zero feedforward / zero a0 panic; 5 ones are fine."""
    c = ctx(be, 2, 512, 44100.0)
    with pytest.raises(waa.WaaError, match="NotSupportedError"):
        c.create_iir_filter([1.0] * 21, [1.0])
    with pytest.raises(waa.WaaError, match="InvalidStateError"):
        c.create_iir_filter([0.0] * 5, [1.0])
    with pytest.raises(waa.WaaError, match="NotSupportedError"):
        c.create_iir_filter([1.0], [1.0] * 21)
    with pytest.raises(waa.WaaError, match="InvalidStateError"):
        c.create_iir_filter([1.0] * 5, [0.0, 1.0, 1.0, 1.0, 1.0])
    with pytest.raises(waa.WaaError, match="NotSupportedError"):
        c.create_iir_filter([], [1.0])
    c.create_iir_filter([1.0] * 5, [1.0] * 5)


def test_c_abi_validation(be):
    """the same panics through the C entry point (not only the Python mirror)"""
    c = ctx(be, 1, 128, 44100.0)
    iir = c.create_iir_filter([1.0], [1.0])
    iir.connect(c.destination())
    c.prepare()
    h = c._handle
    dp = lambda a: np.asarray(a, np.float64).ctypes.data_as(waa.api._DP)
    assert be.iir_set_coefficients(h, iir.id, dp([0.0, 0.0]), 2, dp([1.0]), 1) == 3
    assert b"InvalidStateError" in be.last_error()
    assert be.iir_set_coefficients(h, iir.id, dp([1.0]), 1, dp([0.0]), 1) == 3
    assert be.iir_set_coefficients(h, iir.id, dp([1.0] * 21), 21, dp([1.0]), 1) == 2
    assert be.iir_set_coefficients(h, 0, dp([1.0]), 1, dp([1.0]), 1) != 0  # node 0 is the destination
    c.close()


def test_one_zero_with_feedback_feedforward_different_length(be):
    """iir_filter.rs:524-543: feedforward [0.5, 0.5], feedback [1.] on a unit impulse, abs_all <= 0."""
    sr = 24000.0
    c = ctx(be, 1, 8000, sr)
    src = c.create_buffer_source()
    src.set_buffer(waa.AudioBuffer(np.array([[1.0]], np.float32), sr))
    iir = c.create_iir_filter([0.5, 0.5], [1.0])
    src.connect(iir).connect(c.destination())
    src.start()
    out = c.start_rendering_sync().data[0, 0]
    expected = np.zeros(8000, np.float32)
    expected[:2] = 0.5
    assert np.array_equal(out, expected)


@pytest.mark.parametrize("ftype", sorted(BIQUAD_EQUIVALENTS))
def test_output_against_biquad(be, ftype):
    """iir_filter.rs:545-748: the IIR node with a biquad's coefficients renders what the BiquadFilterNode renders
    (the reference asserts abs_all <= 0 on samples/white.ogg, which is not available offline: seeded white
    noise here, and one f32 ulp of slack because direct form I and transposed form II round differently)."""
    fb, ff = BIQUAD_EQUIVALENTS[ftype]
    noise = white_noise(1, 1, 1000, seed0=1234)
    outs = []
    for kind in ("biquad", "iir"):
        c = ctx(be, 1, 1000, 44100.0)
        if kind == "biquad":
            f = c.create_biquad_filter(type_=ftype, frequency=2000.0, q=1.0, gain=3.0)
        else:
            f = c.create_iir_filter(ff, fb)
        f.connect(c.destination())
        src = c.create_buffer_source()
        src.connect(f)
        src.set_buffer(waa.AudioBuffer(noise[0], 44100.0))
        src.start()
        outs.append(c.start_rendering_sync().data[0, 0])
    a, b = outs
    assert np.max(np.abs(a - b)) <= 2.5e-7
    assert np.mean(a != b) <= 0.02  # nearly every sample is bit-identical


def test_get_frequency_response_scipy_vector(be):
    """iir_filter.rs:750-791 (scipy cheby2 reference, abs_all <= 0 on the magnitudes)"""
    ref_mag = np.float32([1e-3, 4.152_807e-4, 1.460_789_5e-3, 5.051_316e-3, 1.130_323_5e-2, 2.230_340_2e-2,
                          4.311_698e-2, 8.843_45e-2, 2.146_620_2e-1, 6.802_952e-1])
    c = ctx(be, 2, 512, 44100.0)
    iir = c.create_iir_filter([0.019_618_022_238_052_212, -0.036_007_928_102_449_24, 0.019_618_022_238_052_21],
                              [1.0, 1.576_436_200_538_313_7, 0.651_680_173_116_867_3])
    hz = [0.0, 2205.0, 4410.0, 6615.0, 8820.0, 11025.0, 13230.0, 15435.0, 17640.0, 19845.0]
    mag, _ = iir.get_frequency_response(hz)
    assert np.array_equal(mag, ref_mag)


@pytest.mark.parametrize("ftype", sorted(set(BIQUAD_EQUIVALENTS) - {"notch"}))
def test_frequency_responses_against_biquad(be, ftype):
    """iir_filter.rs:793-938: abs_all <= 1e-6 on magnitude and phase (the reference leaves the notch out, :888:
    the phase at the exact notch frequency is the argument of a rounding-sized number)"""
    fb, ff = BIQUAD_EQUIVALENTS[ftype]
    c = ctx(be, 1, 128, 44100.0)
    freqs = [400.0, 800.0, 1200.0, 1600.0, 2000.0, 2400.0, 2800.0, 3200.0, 3600.0, 4000.0]
    bq = c.create_biquad_filter(type_=ftype, frequency=2000.0, q=1.0, gain=3.0)
    iir = c.create_iir_filter(ff, fb)
    m0, p0 = bq.get_frequency_response(freqs)
    m1, p1 = iir.get_frequency_response(freqs)
    assert np.max(np.abs(m0 - m1)) <= 1e-6
    assert np.max(np.abs(p0 - p1)) <= 1e-6


def test_frequency_response_nan_outside_range(be):
    """iir_filter.rs:236-244"""
    c = ctx(be, 1, 128, 44100.0)
    iir = c.create_iir_filter([1.0, 0.5], [1.0, -0.5])
    mag, phase = iir.get_frequency_response([-1.0, 22051.0, 100.0])
    assert np.isnan(mag[:2]).all() and np.isnan(phase[:2]).all() and np.isfinite(mag[2])


def test_render_without_coefficients_is_an_error(be):
    c = ctx(be, 1, 128, 44100.0)
    iir = c.create_iir_filter([1.0], [1.0])
    iir._apply = lambda ctx_: None  # skip the constructor-time upload
    iir.connect(c.destination())
    with pytest.raises(waa.WaaError, match="InvalidStateError"):
        c.start_rendering_sync()


# --------------------------------------------------------------------------- oracle vs scipy (independent check)
@pytest.mark.parametrize("order", [1, 2, 5, 9])
def test_oracle_matches_scipy_lfilter(orc, order):
    """the transposed direct form II of iir_filter.rs is scipy.signal.lfilter's structure: same numbers in f64"""
    b, a = signal.butter(order, 0.2)
    noise = white_noise(1, 1, RQ * 20, seed0=5)
    c = ctx(orc, 1, RQ * 20, 48000.0)
    src = c.create_buffer_source()
    src.set_buffer(waa.AudioBuffer(noise[0], 48000.0))
    iir = c.create_iir_filter(b, a)
    src.connect(iir).connect(c.destination())
    src.start()
    out = c.start_rendering_sync().data[0, 0]
    ref = signal.lfilter(b, a, noise[0, 0].astype(np.float64))
    assert np.max(np.abs(out - ref.astype(np.float32))) <= 1e-7


def test_plan_routes_iir_to_the_streaming_kernel(hip):
    """CPU: plan-only batch; source -> gain -> IIR -> gain -> destination is cut around the IIR"""
    c = waa.OfflineAudioContext(2, RQ * 64, 48000.0, n_instances=4, binding=hip, device=waa.PLAN_ONLY)
    src = c.create_buffer_source()
    src.set_buffer_batch(white_noise(4, 2, RQ * 64), 48000.0)
    b, a = signal.butter(5, 0.1)
    iir = c.create_iir_filter(b, a)
    g0, g1 = c.create_gain(gain=0.5), c.create_gain(gain=0.25)
    src.connect(g0).connect(iir).connect(g1).connect(c.destination())
    src.start()
    plan = c.plan_describe()
    assert "iir_stream states=5" in plan
    assert "IIR" not in plan.replace("iir_stream", "")  # never on the interpreter
    c.close()


@pytest.mark.parametrize("order,wn,kernel", [(2, 0.25, "iir_stream states=2"), (4, 0.25, "iir_stream states=4"),
                                             (11, 0.1, "iir_exact(row) states=11"), (19, 0.1, "iir_exact(row) states=19"),
                                             (6, 0.02, "iir_exact(row) states=6")])
def test_plan_sends_ill_conditioned_filters_to_the_exact_kernel(hip, order, wn, kernel):
    """the lane scan multiplies by powers of the 32-step transition; where those are large (clustered poles)
    the planner picks the lane-per-stream kernel, which needs a materialised input signal"""
    c = waa.OfflineAudioContext(2, RQ * 64, 48000.0, n_instances=4, binding=hip, device=waa.PLAN_ONLY)
    src = c.create_buffer_source()
    src.set_buffer_batch(white_noise(4, 2, RQ * 64), 48000.0)
    b, a = signal.butter(order, wn)
    src.connect(c.create_iir_filter(b, a)).connect(c.destination())
    src.start()
    plan = c.plan_describe()
    assert kernel in plan
    if "exact" in kernel:
        assert "in=signal" in plan.split("iir_exact")[1].splitlines()[0]
    c.close()


def test_plan_unstable_filter_is_exact(hip):
    c = waa.OfflineAudioContext(1, RQ * 64, 48000.0, n_instances=2, binding=hip, device=waa.PLAN_ONLY)
    src = c.create_buffer_source()
    src.set_buffer_batch(white_noise(2, 1, RQ * 64), 48000.0)
    src.connect(c.create_iir_filter([1.0, 0.3], [1.0, -2.5, 1.2])).connect(c.destination())
    src.start()
    assert "iir_exact(lane) states=2" in c.plan_describe()  # low order: one lane per stream
    c.close()


# --------------------------------------------------------------------------- GPU parity
def _render_iir(binding, noise, ff, fb, length=None, sr=48000.0, pre_gain=None, stop_after=None, channels=2):
    n_inst, n_ch, frames = noise.shape
    c = waa.OfflineAudioContext(channels, length or frames, sr, n_instances=n_inst, binding=binding)
    src = c.create_buffer_source()
    src.set_buffer_batch(noise, sr)
    iir = c.create_iir_filter(ff, fb)
    head = src
    if pre_gain is not None:
        head = src.connect(c.create_gain(gain=pre_gain))
    head.connect(iir).connect(c.destination())
    src.start()
    out = c.start_rendering_sync().data
    c.close()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("wn", [0.1, 0.3, 0.6])
@pytest.mark.parametrize("order", [1, 2, 3, 4, 6, 8, 11, 12, 16, 19])
def test_iir_parity_butterworth(hip, orc, order, wn):
    """every kernel size (2, 4, 8, 12, 19 state variables, zero-padded in between) on both kernels (the scan
    kernel for well-conditioned filters, the exact lane kernel for the others), several tiles, 2 channels"""
    b, a = signal.butter(order, wn)
    noise = white_noise(6, 2, 2048 * 3 + 517, seed0=order)
    g = _render_iir(hip, noise, b, a)
    o = _render_iir(orc, noise, b, a)
    assert np.isfinite(o).all()
    scale = max(1.0, float(np.abs(o).max()))  # order 19 at 0.1 is unstable once rounded to f64: huge but finite
    assert rms_err(g, o).max() <= TOL * scale
    assert np.abs(g - o).max() <= 1e-6 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["WAA_IIR_LANE", "WAA_IIR_ROW"])
@pytest.mark.parametrize("order", [1, 2, 5, 10, 16, 17, 19])
def test_iir_exact_kernels_are_bit_identical(hip, orc, order, kernel, monkeypatch):
    """WAA_IIR_EXACT forces an exact kernel (lane per stream / DPP row per stream): the same operations in the
    same order as the oracle"""
    monkeypatch.setenv("WAA_IIR_EXACT", "1")
    monkeypatch.setenv(kernel, "1")
    b, a = signal.butter(order, 0.3)
    noise = white_noise(70, 2, 2048 + 300, seed0=order)  # 140 streams: more than two waves, last one partial
    g = _render_iir(hip, noise, b, a)
    o = _render_iir(orc, noise, b, a)
    assert np.array_equal(g, o)


@pytest.mark.gpu
def test_iir_parity_unequal_lengths_mono_and_tail(hip, orc):
    """feedforward longer than feedback (FIR-like), mono stream, source shorter than the render (tail + silence)"""
    ff = [0.2, -0.1, 0.05, 0.3, 0.1, -0.2]
    fb = [2.0, -0.8]
    noise = white_noise(5, 1, RQ * 9 + 11, seed0=77)
    g = _render_iir(hip, noise, ff, fb, length=2048 * 2 + 100, channels=1)
    o = _render_iir(orc, noise, ff, fb, length=2048 * 2 + 100, channels=1)
    assert rms_err(g, o).max() <= TOL
    assert np.abs(g - o).max() <= 1e-7


@pytest.mark.gpu
def test_iir_parity_behind_gain_and_before_biquad(hip, orc):
    """IIR in the middle of a fused chain: src -> gain -> IIR -> biquad -> gain -> destination"""
    b, a = signal.cheby1(4, 1.0, 0.2)
    noise = white_noise(4, 2, 2048 * 2, seed0=9)
    outs = []
    for be_ in (hip, orc):
        c = waa.OfflineAudioContext(2, 2048 * 2, 48000.0, n_instances=4, binding=be_)
        src = c.create_buffer_source()
        src.set_buffer_batch(noise, 48000.0)
        iir = c.create_iir_filter(b, a)
        bq = c.create_biquad_filter(type_="highpass", frequency=300.0)
        src.connect(c.create_gain(gain=0.7)).connect(iir).connect(bq).connect(c.create_gain(gain=0.5)).connect(c.destination())
        src.start()
        outs.append(c.start_rendering_sync().data)
        c.close()
    assert rms_err(*outs).max() <= TOL


@pytest.mark.gpu
def test_iir_unstable_filter_takes_the_exact_path(hip, orc):
    """poles outside the unit circle: the state overflows, the reference flushes inf/NaN outputs to zero
    (iir_filter.rs:383-385) and recovers; the kernel detects it and replays those tiles serially."""
    ff = [1.0, 0.3]
    fb = [1.0, -2.5, 1.2]
    noise = white_noise(3, 2, 2048 * 2 + 300, seed0=21)
    g = _render_iir(hip, noise, ff, fb)
    o = _render_iir(orc, noise, ff, fb)
    assert np.isfinite(o).all() == np.isfinite(g).all()
    fin = np.isfinite(o) & (np.abs(o) < 1e30)
    assert np.array_equal(np.isfinite(g), np.isfinite(o))
    # before the blow-up both agree tightly; afterwards values are astronomically large and compared relatively
    rel = np.abs(g[fin] - o[fin]) / np.maximum(1.0, np.abs(o[fin]))
    assert rel.max() <= 1e-5
