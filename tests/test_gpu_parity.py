"""Parity of the HIP path against the oracle on seeded inputs (the parity tests proper; -m gpu).

Tolerance: the north star's 1e-6 RMS per channel.  In practice the f64 biquad and the f32
element-wise ops are bit-identical to the oracle; tests assert the tight bound where it holds.
"""
import numpy as np
import pytest

import web_audio_api_rs_amd as waa
from graphs import c2, c5, rms_err, white_noise

pytestmark = pytest.mark.gpu
RQ = 128
TOL = 1e-6
# AnalyserNode spectra, device vs oracle: linear magnitude relative to the row's peak, and the dB bound that follows from it on
# bins within 60 dB of the peak (derivation and the all-contexts measurement: tests/test_full_size_all_instances.py)
ANALYSER_LIN_TOL = 4e-6
ANALYSER_DB_TOL = 0.035


def both(builder, hip, orc, *a, **kw):
    outs = []
    for b in (hip, orc):
        ctx, _ = builder(b, *a, **kw)
        outs.append(ctx.start_rendering_sync().data)
        ctx.close()
    return outs


@pytest.mark.parametrize("length", [RQ * 100 + 37, 2048 * 3, 1, 127, 2049])
def test_c2_small(hip, orc, length):
    noise = white_noise(16, 2, RQ * 120)
    g, o = both(c2, hip, orc, noise, length=length)
    assert g.shape == o.shape == (16, 2, length)
    assert rms_err(g, o).max() <= TOL
    assert np.abs(g - o).max() <= 1e-7  # f64 recurrence: expect (near) bit-identity


@pytest.mark.parametrize("ftype", sorted(waa.BIQUAD_TYPE))
def test_biquad_types_per_instance_params(hip, orc, ftype):
    n = 12
    noise = white_noise(n, 2, RQ * 64)
    rng = np.random.default_rng(11)
    f, q, gdb, det = rng.uniform(30, 20000, n), rng.uniform(0.2, 12, n), rng.uniform(-20, 20, n), rng.uniform(-1200, 1200, n)
    outs = []
    for b in (hip, orc):
        ctx, nodes = c2(b, noise, ftype=ftype)
        for i in range(n):
            nodes["biquad"].frequency.set_value(f[i], instance=i)
            nodes["biquad"].q.set_value(q[i], instance=i)
            nodes["biquad"].gain.set_value(gdb[i], instance=i)
            nodes["biquad"].detune.set_value(det[i] if i % 2 else 0.0, instance=i)
            nodes["gain"].gain.set_value(0.25 + 0.05 * i, instance=i)
        outs.append(ctx.start_rendering_sync().data)
        ctx.close()
    assert rms_err(*outs).max() <= TOL


def test_biquad_mono_and_degenerate_coefficients(hip, orc):
    noise = white_noise(6, 1, RQ * 40)
    cases = [("lowpass", 24000.0, 1.0), ("highpass", 0.0, 1.0), ("bandpass", 1000.0, 0.0), ("notch", 24000.0, 1.0),
             ("allpass", 500.0, 0.0), ("peaking", 0.0, 1.0)]
    outs = []
    for b in (hip, orc):
        ctx, nodes = c2(b, noise)
        for i, (t, f, q) in enumerate(cases):
            nodes["biquad"].frequency.set_value(f, instance=i)
            nodes["biquad"].q.set_value(q, instance=i)
        outs.append(ctx.start_rendering_sync().data)
        ctx.close()
    assert rms_err(*outs).max() <= TOL
    # and each degenerate type on its own
    for t, f, q in cases:
        g, o = both(c2, hip, orc, noise[:2], ftype=t, freq=f, q=q)
        assert rms_err(g, o).max() <= TOL, t


def test_k_rate_and_a_rate_param_blocks(hip, orc):
    """AudioParamValues of len 1 changing per quantum (biquad frequency, gain) and len 128 (gain)."""
    n, nq = 4, 48
    noise = white_noise(n, 2, RQ * nq)
    rng = np.random.default_rng(5)
    fq = np.geomspace(50.0, 12000.0, nq).astype(np.float32)
    ga = rng.uniform(0.1, 1.0, (nq, RQ)).astype(np.float32)
    outs = []
    for b in (hip, orc):
        ctx, nodes = c2(b, noise)
        nodes["biquad"].frequency.set_block(0, fq)
        nodes["biquad"].q.set_block(10, np.linspace(0.5, 5.0, 20).astype(np.float32), instance=1)
        nodes["gain"].gain.set_block(0, ga)
        outs.append(ctx.start_rendering_sync().data)
        ctx.close()
    assert rms_err(*outs).max() <= TOL


def test_gain_fast_paths(hip, orc):
    """gain.rs:163-179: |g| <= 1e-6 mutes, |1-g| <= 1e-6 passes through."""
    noise = white_noise(4, 2, RQ * 8)
    outs = []
    for b in (hip, orc):
        ctx, nodes = c2(b, noise)
        for i, g in enumerate([0.0, 5e-7, 1.0 + 5e-7, 0.999]):
            nodes["gain"].gain.set_value(g, instance=i)
        outs.append(ctx.start_rendering_sync().data)
        ctx.close()
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("rate,buf_sr,loop", [(1.5, None, True), (1.0, 38000.0, True), (0.37, None, False),
                                              (-1.0, None, True), (2.0, 44100.0, False)])
def test_c5_small(hip, orc, rate, buf_sr, loop):
    noise = white_noise(6, 2, 5000)
    g, o = both(c5, hip, orc, noise, length=RQ * 60 + 5, rate=rate, buf_sr=buf_sr, loop=loop)
    assert rms_err(g, o).max() <= TOL
    assert np.abs(g - o).max() <= 1e-6


def test_source_scheduling_per_instance(hip, orc):
    """Distinct start / stop / offset / duration per instance: fast and slow tracks, sub-sample starts."""
    sr, n = 48000.0, 8
    noise = white_noise(n, 2, RQ * 30)
    outs = []
    for b in (hip, orc):
        ctx = waa.OfflineAudioContext(2, RQ * 40, sr, n_instances=n, binding=b)
        src = ctx.create_buffer_source()
        src.set_buffer_batch(noise, sr)
        src.connect(ctx.destination())
        sched = [(0.0, 0.0, None), (RQ * 2 / sr, 0.0, None), (1.5 / sr, 0.0, None), (0.0, 10.25 / sr, None),
                 (300.0 / sr, 0.0, 2000.5 / sr), (0.0, 0.0, 1000.0 / sr), (RQ * 3 / sr, 64.0 / sr, None),
                 (0.01, 0.002, 0.03)]
        for i, (when, off, dur) in enumerate(sched):
            if dur is None:
                src.start_at_with_offset(when, off, instance=i)
            else:
                src.start_at_with_offset_and_duration(when, off, dur, instance=i)
            if i % 3 == 0:
                src.stop_at(when + (2500.3 + i) / sr, instance=i)
        outs.append(ctx.start_rendering_sync().data)
        ctx.close()
    assert np.abs(outs[0] - outs[1]).max() <= 1e-7


@pytest.mark.parametrize("n_ch", [4, 6])
@pytest.mark.parametrize("kind", ["const", "k-rate", "a-rate", "iir"])
def test_filters_on_wide_signals(hip, orc, n_ch, kind):
    """BiquadFilterNode / IIRFilterNode on quad and 5.1 signals: per-channel state of any width (biquad_filter.rs:797-812,
    iir_filter.rs:323-405), one wavefront per (instance, channel) on the streaming kernels — constant, per-quantum and
    per-frame coefficients — then the speakers down-mix to the stereo destination (quantum.rs:399-505)."""
    n_inst, frames = 3, RQ * 50 + 9
    noise = white_noise(n_inst, n_ch, frames)
    outs = []
    for b in (hip, orc):
        ctx = waa.OfflineAudioContext(2, frames, 48000.0, n_instances=n_inst, binding=b)
        src = ctx.create_buffer_source()
        src.set_buffer_batch(noise, 48000.0)
        if kind == "iir":
            from scipy import signal
            bb, aa = signal.butter(4, 0.2)
            flt = ctx.create_iir_filter(bb, aa)
        else:
            flt = ctx.create_biquad_filter(type_="peaking", frequency=900.0, q=2.0, gain=4.0)
            if kind == "k-rate":
                flt.frequency.set_block(0, np.geomspace(200.0, 6000.0, (frames + RQ - 1) // RQ).astype(np.float32))
            elif kind == "a-rate":
                flt.frequency.set_value_at_time(200.0, 0.0)
                flt.frequency.exponential_ramp_to_value_at_time(6000.0, frames / 48000.0)
        src.connect(flt).connect(ctx.create_gain(gain=0.5)).connect(ctx.destination())
        src.start()
        if b is hip:
            assert ("iir_" if kind == "iir" else "biquad_lanes" if kind == "a-rate" else "biquad_stream") in ctx.plan_describe()
        outs.append(ctx.start_rendering_sync().data)
        ctx.close()
    assert rms_err(*outs).max() <= TOL
    assert np.abs(outs[0] - outs[1]).max() <= 2e-6


@pytest.mark.measure
def test_time_parallel_biquad_forms(hip, orc, monkeypatch):
    """Two time-parallel forms of the streaming Biquad against the oracle and against the one-wavefront-per-stream kernel:
    * waa_biquad_scan.hip (WAA_BIQUAD_SCAN=1, constant coefficients): one unit per tile and stream, the incoming state from a
      chained scan over the stream's tiles (self-validating words, no fences) — odd stream counts (the eight unit shards are
      uneven), a source that starts late and ends early (tiles the fast track does not cover, x history through the hand-off);
    * waa_biquad_lanes.hip (default for a-rate params with one table for all instances): one LANE per stream, tile digests,
      two passes — 67 streams (a ragged last group), constant gains behind the filter, a k-rate + a-rate param mix."""
    n_inst, frames = 11, 2048 * 6 + 300
    noise = white_noise(n_inst, 3, 2048 * 5)

    def build(be, arate):
        ctx = waa.OfflineAudioContext(2, frames, 48000.0, n_instances=n_inst, binding=be)
        src = ctx.create_buffer_source()
        src.set_buffer_batch(noise, 48000.0)
        bq = ctx.create_biquad_filter(type_="bandpass", frequency=700.0, q=3.0)
        for i in range(n_inst if not arate else 0):  # (per-instance coefficients; the a-rate form shares ONE table)
            bq.q.set_value(0.5 + i, instance=i)
        if arate:
            bq.frequency.set_value_at_time(90.0, 0.0)
            bq.frequency.exponential_ramp_to_value_at_time(9000.0, frames / 48000.0)
            bq.detune.set_value_at_time(300.0, 0.1)
        src.connect(bq).connect(ctx.create_gain(gain=0.7)).connect(ctx.create_gain(gain=1.0)).connect(ctx.destination())
        src.start_at(0.0 if arate else 77 / 48000.0)
        return ctx

    for arate in (False, True):
        octx = build(orc, arate)
        ref = octx.start_rendering_sync().data
        octx.close()
        outs = {}
        for env in ({}, {"WAA_BIQUAD_SCAN": "1"}, {"WAA_ARATE_STREAM": "1"}):
            for k in ("WAA_BIQUAD_SCAN", "WAA_ARATE_STREAM"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            ctx = build(hip, arate)
            plan = ctx.plan_describe()
            outs[tuple(env)] = ctx.start_rendering_sync().data
            ctx.close()
            if arate:
                assert ("biquad_lanes" in plan) == ("WAA_ARATE_STREAM" not in env), plan
            assert rms_err(outs[tuple(env)], ref).max() <= TOL, (arate, env)
            assert np.abs(outs[tuple(env)] - ref).max() <= 2e-6
        assert np.abs(outs[()] - outs[("WAA_ARATE_STREAM",)]).max() <= 1e-6


def test_panners_and_mixing(hip, orc):
    """mono/stereo StereoPanner, equal-power Panner, fan-in summing, mono -> stereo destination."""
    sr, n = 44100.0, 4
    for nch in (1, 2):
        noise = white_noise(n, nch, RQ * 20)
        outs = []
        for b in (hip, orc):
            ctx = waa.OfflineAudioContext(2, RQ * 20, sr, n_instances=n, binding=b)
            src = ctx.create_buffer_source()
            src.set_buffer_batch(noise, sr)
            sp = ctx.create_stereo_panner()
            for i, p in enumerate([-1.0, -0.3, 0.1, 1.0]):
                sp.pan.set_value(p, instance=i)
            pn = ctx.create_panner(distance_model="inverse", ref_distance=1.0, rolloff_factor=1.0,
                                   cone_inner_angle=60.0, cone_outer_angle=120.0, cone_outer_gain=0.3)
            for i in range(n):
                pn.position_x.set_value(-2.0 + 1.5 * i, instance=i)
                pn.position_z.set_value(-1.0 - 0.3 * i, instance=i)
                pn.position_y.set_value(0.2 * i, instance=i)
            k = ctx.create_constant_source(offset=0.125)
            g = ctx.create_gain(gain=0.7)
            src.connect(sp).connect(g)
            src.connect(pn).connect(g)
            k.connect(g)
            g.connect(ctx.destination())
            src.start()
            k.start_at(RQ * 3.5 / sr)
            k.stop_at(RQ * 9.25 / sr)
            outs.append(ctx.start_rendering_sync().data)
            ctx.close()
        assert rms_err(*outs).max() <= TOL, nch
        assert np.abs(outs[0] - outs[1]).max() <= 1e-6


def test_pan_automation_blocks(hip, orc):
    sr, n, nq = 48000.0, 3, 16
    noise = white_noise(n, 2, RQ * nq)
    rng = np.random.default_rng(9)
    pk = rng.uniform(-1, 1, nq).astype(np.float32)
    pa = rng.uniform(-1, 1, (nq, RQ)).astype(np.float32)
    for blocks in (pk, pa):
        outs = []
        for b in (hip, orc):
            ctx = waa.OfflineAudioContext(2, RQ * nq, sr, n_instances=n, binding=b)
            src = ctx.create_buffer_source()
            src.set_buffer_batch(noise, sr)
            sp = ctx.create_stereo_panner()
            sp.pan.set_block(0, blocks)
            src.connect(sp).connect(ctx.destination())
            src.start()
            outs.append(ctx.start_rendering_sync().data)
            ctx.close()
        # a-rate: device sinf vs libm sinf may differ by 1 ulp
        assert rms_err(*outs).max() <= TOL


def test_c2_full_size_sampled_instances(hip, orc):
    """BASELINE config C2 at full size (1024 contexts x 10 s); the oracle renders a sample of instances."""
    n_inst, frames = 1024, 480000
    rng = np.random.default_rng(123)
    noise = rng.uniform(-1.0, 1.0, (n_inst, 2, frames)).astype(np.float32)
    ctx, _ = c2(hip, noise)
    out = ctx.start_rendering_sync().data
    ctx.close()
    pick = [0, 1, 511, 1023]
    octx, _ = c2(orc, noise[pick])
    ref = octx.start_rendering_sync().data
    octx.close()
    assert rms_err(out[pick], ref).max() <= TOL
    assert np.abs(out[pick] - ref).max() <= 1e-7
    # size-independent property: linearity in the input (gain commutes) on two other instances
    assert np.all(np.isfinite(out)) and out.shape == (n_inst, 2, frames)


# ----------------------------------------------------------------------------- convolver / analyser
from graphs import c4, garage_ir, garage_like_ir, t1  # noqa: E402


def _decaying_ir(n_ch, frames, seed=3, tau=0.3):
    rng = np.random.default_rng(seed)
    t = np.arange(frames) / max(frames, 1)
    return (rng.uniform(-1, 1, (n_ch, frames)) * np.exp(-t / tau)).astype(np.float32)


@pytest.mark.parametrize("ir_len,n_inst", [(1, 3), (300, 5), (3000, 4), (3073, 2), (20000, 5), (70000, 3)])
def test_t1_small_multi_partition(hip, orc, ir_len, n_inst):
    """src -> Biquad -> Convolver -> destination; IRs from 1 tap to 69 reference partitions; odd batch
    sizes exercise the instance pairing.  Oracle = restated fft-convolver (f32, 1024-frame partitions)."""
    frames = RQ * 96 + 11
    noise = white_noise(n_inst, 2, frames)
    ir = _decaying_ir(2, ir_len)
    g, o = both(t1, hip, orc, noise, ir)
    assert rms_err(g, o).max() <= TOL
    assert np.abs(g - o).max() <= 2e-6


@pytest.mark.parametrize("in_ch,ir_ch", [
    (1, 1), (1, 2), (2, 2), (2, 4), (1, 4),
    (2, 1),  # after the stereo source has ended the reference falls back to the (1,1) routing: convolver 0's tail in
             # both channels (convolver.rs:343-392) — rendered by the dynamic-count path (waa_dyn.hip), was a strict xfail
])
def test_convolver_channel_configs_random(hip, orc, in_ch, ir_ch):
    """convolver.rs:384-466 routing with random data, normalisation on, plus a tail after the source ends."""
    n_inst, frames = 4, RQ * 40
    noise = white_noise(n_inst, in_ch, RQ * 25)  # the source ends before the render does: tail time
    ir = _decaying_ir(ir_ch, 1500, seed=in_ch * 10 + ir_ch)
    g, o = both(t1, hip, orc, noise, ir, length=frames, with_biquad=False)
    assert rms_err(g, o).max() <= TOL


def test_convolver_exact_linear_convolution(hip, orc_lib):
    """Anchor on the mathematical definition (f64 direct convolution), not only on the oracle's f32 FFTs."""
    import ctypes as C
    frames, ir_len = RQ * 64, 5000
    noise = white_noise(2, 1, frames)
    ir = _decaying_ir(1, ir_len)
    ctx, _ = t1(hip, noise, ir, with_biquad=False)
    out = ctx.start_rendering_sync().data
    ctx.close()
    # normalisation scale as the reference computes it (convolver.rs:16-53), via the oracle helper
    FP, DP = C.POINTER(C.c_float), C.POINTER(C.c_double)
    orc_lib.orc_convolver_normalization_scale.restype = C.c_float
    orc_lib.orc_convolver_normalization_scale.argtypes = [C.POINTER(FP), C.c_uint32, C.c_uint64, C.c_float]
    chans = (FP * 1)(ir[0].ctypes.data_as(FP))
    scale = orc_lib.orc_convolver_normalization_scale(chans, 1, ir_len, 48000.0)
    h = (ir[0] * np.float32(scale)).astype(np.float32)
    orc_lib.orc_convolve_exact.argtypes = [FP, C.c_uint64, FP, C.c_uint64, DP, C.c_uint64]
    for i in range(2):
        ye = np.zeros(frames, np.float64)
        x = np.ascontiguousarray(noise[i, 0])
        orc_lib.orc_convolve_exact(x.ctypes.data_as(FP), frames, h.ctypes.data_as(FP), ir_len, ye.ctypes.data_as(DP), frames)
        # mono source + mono IR -> mono output, up-mixed to both destination channels
        for c in range(2):
            assert np.sqrt(np.mean((out[i, c] - ye) ** 2)) <= TOL


def test_garage_sized_ir_sampled(hip, orc):
    """C3/T1 IR shape (2 ch x 178 899 frames = 175 reference partitions; 22 blocks of 8192 on the device)."""
    n_inst, frames = 6, RQ * 300
    noise = white_noise(n_inst, 2, frames)
    ir = garage_like_ir()  # (synthetic stand-in with a hand-set noise floor; the real IR is used by the tests below)
    ctx, _ = t1(hip, noise, ir)
    out = ctx.start_rendering_sync().data
    ctx.close()
    pick = [0, 5]
    octx, _ = t1(orc, noise[pick], ir)
    ref = octx.start_rendering_sync().data
    octx.close()
    assert rms_err(out[pick], ref).max() <= TOL


def test_c4_small_with_analyser(hip, orc):
    n_inst, frames = 4, RQ * 64
    noise = white_noise(n_inst, 2, frames)
    ir = _decaying_ir(2, 4000)
    res = []
    for b in (hip, orc):
        ctx, nodes = c4(b, noise, ir)
        out = ctx.start_rendering_sync().data
        an = nodes["analyser"]
        res.append((out, [an.get_float_frequency_data(instance=i) for i in range(n_inst)],
                    [an.get_float_time_domain_data(instance=i) for i in range(n_inst)],
                    [an.get_byte_frequency_data(instance=i) for i in range(n_inst)],
                    [an.get_byte_time_domain_data(n=100, instance=i) for i in range(n_inst)]))
        ctx.close()
    (g, gf, gt, gbf, gbt), (o, of, ot, obf, obt) = res
    assert rms_err(g, o).max() <= TOL
    for i in range(n_inst):
        assert np.abs(gt[i] - ot[i]).max() <= 2e-6
        # dB of a 2048-bin f32 FFT: bins at the f32 noise floor (-200 dB after the 200 Hz lowpass) are rounding
        # noise in both implementations, so compare linear magnitudes
        gl, ol = 10.0 ** (gf[i].astype(np.float64) / 20), 10.0 ** (of[i].astype(np.float64) / 20)
        # two f32 transforms of 2048 points over the same samples: a few 1e-7 of the row's peak each (the tolerance and the
        # dB bound derived from it: tests/test_full_size_all_instances.py, where all 512 contexts of C4 are compared)
        print(f"analyser instance {i}: linear diff / peak {np.abs(gl - ol).max() / ol.max():.3e}")
        assert np.abs(gl - ol).max() <= ANALYSER_LIN_TOL * np.abs(ol).max()
        big = ol > 1e-3 * ol.max()
        assert np.abs(gf[i][big] - of[i][big]).max() <= ANALYSER_DB_TOL
        assert np.abs(gbf[i].astype(int) - obf[i].astype(int)).max() <= 1
        assert np.abs(gbt[i].astype(int) - obt[i].astype(int)).max() <= 1


@pytest.mark.parametrize("fft_size", [32, 64, 128, 1024, 4096, 32768])
def test_analyser_fft_sizes(hip, orc, fft_size):
    sr, frames = 48000.0, RQ * 300
    noise = white_noise(2, 2, frames)
    res = []
    for b in (hip, orc):
        ctx = waa.OfflineAudioContext(2, frames, sr, n_instances=2, binding=b)
        src = ctx.create_buffer_source()
        src.set_buffer_batch(noise, sr)
        an = ctx.create_analyser(fft_size=fft_size, smoothing_time_constant=0.3)
        src.connect(an).connect(ctx.destination())
        src.start()
        ctx.start_rendering_sync()
        res.append((an.get_float_frequency_data(instance=1), an.get_float_time_domain_data(instance=1)))
        ctx.close()
    assert np.array_equal(res[0][1], res[1][1])
    gl, ol = 10.0 ** (res[0][0].astype(np.float64) / 20), 10.0 ** (res[1][0].astype(np.float64) / 20)
    assert np.abs(gl - ol).max() <= 1e-6 * max(1.0, np.abs(ol).max())


def test_many_sources_fan_in(hip, orc):
    """graph.rs:524-535 — 9 sources summed into the destination (fan-in above the kernel's 4 inputs)."""
    sr = 44100.0
    d = np.zeros((2, 512), np.float32)
    d[:, 0] = 1.0
    d[0, 5] = 0.25
    outs = []
    for b in (hip, orc):
        ctx = waa.OfflineAudioContext(2, 44100, sr, binding=b)
        for idx in [0, 3, 512, 517, 1000, 1005, 20000, 21234, 37590]:
            s = ctx.create_buffer_source()
            s.set_buffer(waa.AudioBuffer(d * (1 + idx % 7), sr))
            s.connect(ctx.destination())
            s.start_at(idx / sr)
        outs.append(ctx.start_rendering_sync().data)
        ctx.close()
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("ftype", ["lowpass", "peaking", "highshelf", "bandpass"])
def test_c1_a_rate_biquad(hip, orc, ftype):
    """BASELINE config C1, a-rate variant (examples/biquad.rs:39-42): frequency exponential ramp 10 Hz -> 10 kHz
    over the render => per-sample coefficients (biquad_filter.rs:837-855). Coefficients come from device
    sin/cos/pow (<= 1-2 ulp from the host libm the oracle uses)."""
    n, frames, sr = 3, RQ * 150 + 9, 48000.0
    noise = white_noise(n, 2, frames)
    outs = []
    for b in (hip, orc):
        ctx, nodes = c2(b, noise, ftype=ftype)
        f = nodes["biquad"].frequency
        f.set_value_at_time(10.0, 0.0)
        f.exponential_ramp_to_value_at_time(10000.0, frames / sr)
        nodes["biquad"].gain.set_value(6.0)
        nodes["biquad"].detune.set_block(20, np.linspace(-600, 600, 30 * RQ).astype(np.float32).reshape(30, RQ), instance=1)
        outs.append(ctx.start_rendering_sync().data)
        ctx.close()
    assert rms_err(*outs).max() <= TOL
    assert np.abs(outs[0] - outs[1]).max() <= 2e-6


def test_t1_full_length_sampled_instances(hip, orc):
    """North-star headline graph (T1: src -> Biquad -> Convolver(garage-sized IR) -> destination) at the full
    10 s render length (59 blocks of 8192, 22 partitions); the oracle renders a sample of the instances."""
    n_inst, frames = 33, 480000  # odd count: the last instance pair is half empty
    noise = white_noise(n_inst, 2, frames)
    ir = garage_ir(hip)  # the reference's parking-garage response, decoded + resampled to 48 kHz: 2 x 178 899
    ctx, _ = t1(hip, noise, ir)
    out = ctx.start_rendering_sync().data
    ctx.close()
    pick = [0, 17, 32]
    octx, _ = t1(orc, noise[pick], ir)
    ref = octx.start_rendering_sync().data
    octx.close()
    err = rms_err(out[pick], ref)
    assert err.max() <= TOL, err
    # size-independent property: time invariance / causality — the first 100 ms do not depend on later input
    noise2 = noise[:2].copy()
    noise2[:, :, 4800:] = 0.0
    ctx, _ = t1(hip, noise2, ir)
    out2 = ctx.start_rendering_sync().data
    ctx.close()
    assert np.abs(out2[:, :, :4800] - out[:2, :, :4800]).max() <= 1e-6


def test_split_chains_biquads_with_elementwise_ops(hip, orc):
    """Planner: constant-coefficient biquads go to the streaming kernel, the ops around them to the element-wise
    kernel (src -> gain -> biquad -> gain -> stereo panner -> biquad -> waveshaper -> destination, mono source)."""
    sr, n = 48000.0, 5
    for nch in (1, 2):
        noise = white_noise(n, nch, RQ * 70 + 3)
        outs = []
        for b in (hip, orc):
            ctx = waa.OfflineAudioContext(2, RQ * 80, sr, n_instances=n, binding=b)
            src = ctx.create_buffer_source()
            src.set_buffer_batch(noise, sr)
            g0 = ctx.create_gain(gain=0.9)
            b1 = ctx.create_biquad_filter(type_="peaking", frequency=1200.0, q=2.0, gain=5.0)
            g1 = ctx.create_gain(gain=0.7)
            pan = ctx.create_stereo_panner(pan=-0.35)
            b2 = ctx.create_biquad_filter(type_="highpass", frequency=90.0, q=0.7)
            i = np.arange(257, dtype=np.float32)
            sh = ctx.create_wave_shaper(curve=np.tanh((i - 128) / 64).astype(np.float32))
            src.connect(g0).connect(b1).connect(g1).connect(pan).connect(b2).connect(sh).connect(ctx.destination())
            for k in range(n):
                b1.frequency.set_value(300.0 + 400.0 * k, instance=k)
            src.start_at(RQ * 2 / sr)
            outs.append(ctx.start_rendering_sync().data)
            ctx.close()
        assert rms_err(*outs).max() <= TOL, nch
        assert np.abs(outs[0] - outs[1]).max() <= 1e-6


def test_k_rate_biquad_streaming_kernel(hip, orc):
    """Per-quantum coefficient changes (k-rate automation of all four params, distinct per instance) on the
    streaming kernel: per-lane matrices + general scan; includes degenerate coefficient sets mid-render."""
    n, nq = 6, 70
    noise = white_noise(n, 2, RQ * nq + 5)
    rng = np.random.default_rng(21)
    qv = rng.uniform(0.3, 8.0, (n, nq - 10)).astype(np.float32)
    gv = rng.uniform(-12, 12, (n, nq)).astype(np.float32)
    dv = rng.uniform(-300, 300, (n, 20)).astype(np.float32)
    for ftype in ("lowpass", "peaking", "notch", "highshelf"):
        outs = []
        for b in (hip, orc):
            ctx, nodes = c2(b, noise, ftype=ftype)
            bq = nodes["biquad"]
            for i in range(n):
                f = np.geomspace(40.0 + 30 * i, 15000.0, nq).astype(np.float32)
                f[nq // 2] = 24000.0 if i % 2 else 0.0  # degenerate sets (f == nyquist / 0) in the middle
                bq.frequency.set_block(0, f, instance=i)
                bq.q.set_block(5, qv[i], instance=i)
                bq.gain.set_block(0, gv[i], instance=i)
                bq.detune.set_block(20, dv[i], instance=i)
            outs.append(ctx.start_rendering_sync().data)
            ctx.close()
        assert rms_err(*outs).max() <= TOL, ftype
        assert np.abs(outs[0] - outs[1]).max() <= 1e-6, ftype


# ----------------------------------------------------------------------------- BASELINE configs at FULL size
def _noise_fast(n_inst, n_ch, frames, seed):
    """uniform white noise in [-1, 1) for a whole batch from one generator (fast enough for GBs)"""
    rng = np.random.default_rng(seed)
    out = rng.random((n_inst, n_ch, frames), dtype=np.float32)
    out *= 2.0
    out -= 1.0
    return out


def test_c3_full_size_real_ir_sampled(hip, orc):
    """BASELINE config 3 at full size: 512 contexts x 10 s, BufferSource -> Convolver(parking-garage IR, normalised)
    -> destination; the oracle (restated fft-convolver, 175 partitions of 1024) renders a sample of the instances."""
    n_inst, frames = 512, 480000
    noise = _noise_fast(n_inst, 2, frames, 31)
    ir = garage_ir(hip)
    assert ir.shape == (2, 178899)
    pick = [0, 1, 255, 510, 511]
    ctx, _ = t1(hip, noise, ir, with_biquad=False)
    assert "P=22 blocks=59" in ctx.plan_describe()
    out = ctx.render_instances(pick)
    ctx.close()
    octx, _ = t1(orc, noise[pick], garage_ir(orc), with_biquad=False)
    ref = octx.start_rendering_sync().data
    octx.close()
    err = rms_err(out, ref)
    assert err.max() <= TOL, err
    assert float(np.abs(ref).max()) > 1e-2


def test_c4_full_size_real_ir_sampled(hip, orc):
    """BASELINE config 4, one GPU's shard at full size: 512 contexts x 10 s, BufferSource -> Biquad -> Convolver(garage
    IR) -> StereoPanner(0.1) -> Analyser(2048, 0.8) -> destination, one get_float_frequency_data pull per sampled
    context after the render."""
    n_inst, frames = 512, 480000
    noise = _noise_fast(n_inst, 2, frames, 32)
    pick = [0, 3, 256, 511]
    ctx, nodes = c4(hip, noise, garage_ir(hip))
    out = ctx.render_instances(pick)
    gf = [nodes["analyser"].get_float_frequency_data(instance=i) for i in pick]
    gt = [nodes["analyser"].get_float_time_domain_data(instance=i) for i in pick]
    ctx.close()
    octx, onodes = c4(orc, noise[pick], garage_ir(orc))
    ref = octx.start_rendering_sync().data
    of = [onodes["analyser"].get_float_frequency_data(instance=i) for i in range(len(pick))]
    ot = [onodes["analyser"].get_float_time_domain_data(instance=i) for i in range(len(pick))]
    octx.close()
    err = rms_err(out, ref)
    assert err.max() <= TOL, err
    for k in range(len(pick)):
        assert np.abs(gt[k] - ot[k]).max() <= 2e-6
        gl, ol = 10.0 ** (gf[k].astype(np.float64) / 20), 10.0 ** (of[k].astype(np.float64) / 20)
        assert np.abs(gl - ol).max() <= ANALYSER_LIN_TOL * np.abs(ol).max()


def _t1_like(be, noise, ir, length, highpass=False, via_gain=False):
    """source [-> Gain with a second consumer] -> Biquad -> Convolver -> destination"""
    n_inst = noise.shape[0]
    ctx = waa.OfflineAudioContext(2, length, 48000.0, n_instances=n_inst, binding=be)
    src = ctx.create_buffer_source()
    src.set_buffer_batch(noise, 48000.0)
    node = src
    if via_gain:  # the Biquad's input is a signal some other consumer materialises anyway (fan-out)
        node = src.connect(ctx.create_gain(gain=0.7))
        node.connect(ctx.create_gain(gain=0.1)).connect(ctx.destination())
    bq = ctx.create_biquad_filter(type_="highpass" if highpass else "lowpass", frequency=900.0 if highpass else 200.0, q=1.3)
    for i in range(n_inst):  # per-instance coefficients
        bq.frequency.set_value(200.0 + 150.0 * i, instance=i)
    node.connect(bq).connect(ctx.create_convolver(buffer=waa.AudioBuffer(ir, 48000.0))).connect(ctx.destination())
    src.start()
    return ctx


@pytest.mark.measure
@pytest.mark.parametrize("case", ["stereo-odd", "mono", "long-buffer", "via-gain", "highpass-one"])
def test_biquad_folded_into_the_forward_transform(hip, orc, monkeypatch, case):
    """source -> Biquad(constant coefficients) -> Convolver(long IR): the forward transform's input stage filters the
    blocks (conv_fft3_fwd_bq_kernel, no Biquad launch, no filtered signal in HBM).  Checked against the oracle and against
    the two-launch plan (WAA_NO_CONV_BIQUAD_FOLD=1): odd instance count (half-empty last pair), mono input, a buffer longer
    than the render, a Biquad fed by a materialised signal, one instance.  (A buffer that ENDS mid-render changes the
    reference's channel counts — the graph then takes the exact per-quantum path, which never folds.)"""
    monkeypatch.setenv("WAA_NO_CONV_BIQUAD_IR_FOLD", "1")  # (one context = "the same coefficients on every context": round 5's fold)
    ir = garage_ir(hip)
    length = 8192 * 4 + 1024  # (a source is read in place when its buffer is a whole number of render quanta)
    n_inst, n_ch, buf = {"stereo-odd": (5, 2, length), "mono": (4, 1, length), "long-buffer": (3, 2, length + 128 * 40),
                         "via-gain": (3, 2, length), "highpass-one": (1, 2, length)}[case]
    noise = white_noise(n_inst, n_ch, buf)
    kw = dict(highpass=case == "highpass-one", via_gain=case == "via-gain")
    ctx = _t1_like(hip, noise, ir, length, **kw)
    plan = ctx.plan_describe()
    assert "filtered by the forward transform's input stage" in plan and "biquad_stream" not in plan
    got = ctx.start_rendering_sync().data
    ctx.close()
    octx = _t1_like(orc, noise, garage_ir(orc), length, **kw)
    ref = octx.start_rendering_sync().data
    octx.close()
    assert rms_err(got, ref).max() <= TOL
    monkeypatch.setenv("WAA_NO_CONV_BIQUAD_FOLD", "1")
    ctx = _t1_like(hip, noise, ir, length, **kw)
    assert "biquad_stream" in ctx.plan_describe()
    two = ctx.start_rendering_sync().data
    ctx.close()
    # same filter arithmetic in the same order; the transforms differ in rounding (pass 1 of the folded kernel multiplies
    # half of its twiddles together, ~1 ulp): a few f32 ulps of the peak
    assert np.abs(got - two).max() <= 1e-6 * max(1.0, float(np.abs(two).max()))


@pytest.mark.measure
@pytest.mark.parametrize("case", ["lowpass-odd", "mono-highpass", "peaking-long-buffer", "notch-via-gain", "allpass-one", "lowshelf-true-stereo"])
def test_biquad_folded_into_the_impulse_response(hip, orc, monkeypatch, case):
    """source -> Biquad(the same constant coefficients on every context) -> Convolver(long IR): Biquad and Convolver are LTI,
    the filter's transfer function is folded into the impulse response at plan time (conv_fold_biquad_into_ir) and the render
    is the plain forward transform.  Against the oracle's sample-by-sample cascade (f64 Biquad, then the restated
    fft-convolver) at the north star's 1e-6 RMS per channel, and against the exact-order filter stage of round 3
    (WAA_NO_CONV_BIQUAD_IR_FOLD=1): every filter type whose ringing dies inside the extension, odd instance count, mono input,
    a buffer longer than the render, a Biquad fed by a materialised signal, a 4-channel (true stereo) response."""
    length = 8192 * 4 + 1024
    type_, freq, q, gain, n_inst, n_ch, buf, via_gain = {
        "lowpass-odd": ("lowpass", 200.0, 1.0, 0.0, 5, 2, length, False),
        "mono-highpass": ("highpass", 900.0, 1.3, 0.0, 4, 1, length, False),
        "peaking-long-buffer": ("peaking", 1500.0, 2.0, 9.0, 3, 2, length + 128 * 40, False),
        "notch-via-gain": ("notch", 3000.0, 4.0, 0.0, 3, 2, length, True),
        "allpass-one": ("allpass", 700.0, 0.8, 0.0, 1, 2, length, False),
        "lowshelf-true-stereo": ("lowshelf", 400.0, 1.0, -6.0, 2, 2, length, False)}[case]
    noise = white_noise(n_inst, n_ch, buf)

    def build(be):
        ir = garage_ir(be)
        if case == "lowshelf-true-stereo":  # four response channels (convolver.rs:419-485)
            ir = np.ascontiguousarray(np.concatenate([ir, ir[::-1] * np.float32(0.5)]))
        ctx = waa.OfflineAudioContext(2, length, 48000.0, n_instances=n_inst, binding=be)
        src = ctx.create_buffer_source()
        src.set_buffer_batch(noise, 48000.0)
        node = src
        if via_gain:
            node = src.connect(ctx.create_gain(gain=0.7))
            node.connect(ctx.create_gain(gain=0.1)).connect(ctx.destination())
        bq = ctx.create_biquad_filter(type_=type_, frequency=freq, q=q, gain=gain)
        node.connect(bq).connect(ctx.create_convolver(buffer=waa.AudioBuffer(ir, 48000.0))).connect(ctx.destination())
        src.start()
        return ctx

    ctx = build(hip)
    plan = ctx.plan_describe()
    assert "folded into the impulse response" in plan and "in the impulse response)" in plan and "biquad_stream" not in plan
    assert "forward transform's input stage" not in plan
    got = ctx.start_rendering_sync().data
    ctx.close()
    octx = build(orc)
    ref = octx.start_rendering_sync().data
    octx.close()
    assert np.isfinite(got).all() and np.abs(ref).max() > 1e-3
    err = rms_err(got, ref)
    print(f"{case}: response fold vs the oracle's cascade: worst per-channel RMS {err.max():.3e} (signal RMS {np.sqrt(np.mean(ref.astype(np.float64) ** 2)):.3e})")
    assert err.max() <= TOL
    monkeypatch.setenv("WAA_NO_CONV_BIQUAD_IR_FOLD", "1")
    ctx = build(hip)
    assert "forward transform's input stage" in ctx.plan_describe()
    kern = ctx.start_rendering_sync().data
    ctx.close()
    assert rms_err(got, kern).max() <= TOL and rms_err(kern, ref).max() <= TOL


@pytest.mark.parametrize("with_biquad", [False, True])
def test_shared_audio_buffer_in_front_of_the_convolver(hip, orc, with_biquad):
    """One AudioBuffer shared by every context (set_buffer for all instances: instance stride 0), read in place by the
    forward transform — with and without the Biquad folded in; per-instance filter coefficients keep the contexts apart."""
    n_inst, length = 3, 8192 * 3 + 128
    noise = white_noise(1, 2, length)[0]
    outs = []
    for be in (hip, orc):
        ctx = waa.OfflineAudioContext(2, length, 48000.0, n_instances=n_inst, binding=be)
        src = ctx.create_buffer_source()
        src.set_buffer(waa.AudioBuffer(noise, 48000.0))
        node = src
        if with_biquad:
            bq = ctx.create_biquad_filter(type_="lowpass", frequency=300.0, q=1.0)
            for i in range(n_inst):
                bq.frequency.set_value(300.0 + 400.0 * i, instance=i)
            node = src.connect(bq)
        node.connect(ctx.create_convolver(buffer=waa.AudioBuffer(garage_ir(be), 48000.0))).connect(ctx.destination())
        src.start()
        if be is hip:
            assert "in place" in ctx.plan_describe()
        outs.append(ctx.start_rendering_sync().data)
        ctx.close()
    assert rms_err(*outs).max() <= TOL
    if with_biquad:
        assert np.abs(outs[0][0] - outs[0][2]).max() > 1e-4  # (the contexts differ)


def test_t1_north_star_size_real_ir_sampled(hip, orc):
    """The north-star target graph at ITS size: 1024 contexts x 10 s, BufferSource -> Biquad(lowpass 200 Hz, Q 1) ->
    Convolver(the real parking-garage response, 2 x 178 899 frames, normalised) -> destination — the batch bench.py's `t1`
    record times (512 instance pairs x 2 channels x 59 blocks through the N = 16384 transforms).  The oracle renders a
    sample of the contexts: first and last, both members of a pair, and the two around the middle."""
    n_inst, frames = 1024, 480000
    noise = _noise_fast(n_inst, 2, frames, 34)
    pick = [0, 1, 511, 512, 1022, 1023]
    ctx, _ = t1(hip, noise, garage_ir(hip))
    plan = ctx.plan_describe()
    assert "P=22 blocks=59 pairs=512" in plan and "the Biquad in front, in the impulse response" in plan
    out = ctx.render_instances(pick)
    ctx.close()
    octx, _ = t1(orc, noise[pick], garage_ir(orc))
    ref = octx.start_rendering_sync().data
    octx.close()
    err = rms_err(out, ref)
    assert err.max() <= TOL, err
    assert float(np.abs(ref).max()) > 1e-3
    # size-independent: the two members of a pair share one complex transform (a + i b) — their results must not leak into
    # each other: instance 0 rendered again next to a DIFFERENT partner gives the same samples
    swapped = noise[[0, 700]].copy()
    ctx, _ = t1(hip, swapped, garage_ir(hip))
    again = ctx.start_rendering_sync().data
    ctx.close()
    assert np.abs(again[0] - out[0]).max() <= 2e-6


@pytest.mark.parametrize("rate,buf_sr,n_ch", [(1.5, None, 2), (1.0, 38000.0, 2), (1.5, None, 1)])
def test_c5_full_size_sampled(hip, orc, rate, buf_sr, n_ch):
    """BASELINE config 5 at full size: 2048 contexts x 10 s, looping BufferSource with playbackRate 1.5 (or a 38 kHz
    buffer in a 48 kHz context) -> WaveShaper(2048-point curve) -> destination; 1e-6 RMS per channel against the
    oracle for a sample of the instances (the 10 s render wraps the 65 536-frame loop 7-11 times)."""
    n_inst, frames, buf_frames = 2048, 480000, 65536
    noise = _noise_fast(n_inst, n_ch, buf_frames, 33)
    pick = [0, 1, 1023, 2046, 2047]
    ctx, _ = c5(hip, noise, length=frames, rate=rate, buf_sr=buf_sr, loop=True)
    out = ctx.render_instances(pick)
    ctx.close()
    octx, _ = c5(orc, noise[pick], length=frames, rate=rate, buf_sr=buf_sr, loop=True)
    ref = octx.start_rendering_sync().data
    octx.close()
    err = rms_err(out, ref)
    assert err.max() <= TOL, err
    assert np.abs(out - ref).max() <= 1e-6


@pytest.mark.parametrize("nch", [1, 2])
def test_audio_rate_listener_automation(hip, orc, nch):
    """panner.rs:830-897, the branch taken when a listener param is a 128-value slice: every frame has its own
    geometry (a-rate listener positionX ramp, a-rate panner positionZ, a k-rate forward vector change halfway); quanta
    in which the listener is single-valued (before the ramp starts) keep the once-per-quantum rule.  Geometry runs on
    the device (waa_panner.hip: device acosf / sinf / cosf, <= 1-2 ulp from the host libm of the oracle)."""
    sr, n, frames = 48000.0, 3, RQ * 60 + 11
    noise = white_noise(n, nch, frames, seed0=17)
    outs = []
    for b in (hip, orc):
        ctx = waa.OfflineAudioContext(2, frames, sr, n_instances=n, binding=b)
        src = ctx.create_buffer_source()
        src.set_buffer_batch(noise, sr)
        pn = ctx.create_panner(distance_model="inverse", ref_distance=1.0, rolloff_factor=1.5, position=(1.0, 0.5, -2.0),
                               cone_inner_angle=40.0, cone_outer_angle=100.0, cone_outer_gain=0.2)
        li = ctx.listener()
        li.position_x.set_value_at_time(-3.0, RQ * 10 / sr).linear_ramp_to_value_at_time(4.0, RQ * 50 / sr)
        li.forward_x.set_value_at_time(0.5, RQ * 30 / sr)
        pn.position_z.set_value_at_time(-2.0, 0.0).linear_ramp_to_value_at_time(1.5, frames / sr)
        for i in range(n):
            pn.position_y.set_value(0.5 - 0.4 * i, instance=i)
        src.connect(pn).connect(ctx.destination())
        src.start()
        if b is hip:
            assert "audio-rate AudioListener automation" in ctx.plan_describe()
        outs.append(ctx.start_rendering_sync().data)
        ctx.close()
    assert rms_err(*outs).max() <= TOL
    assert np.abs(outs[0] - outs[1]).max() <= 5e-6


@pytest.mark.parametrize("listener_too", [False, True])
def test_per_instance_panner_automation(hip, orc, listener_too):
    """Per-instance event lists on an equal-power PannerNode's OWN params with a single-valued listener: the geometry is
    evaluated once per quantum from the first value of every param (panner.rs:833-846) — host work; only AudioListener
    params are replayed on the device.  With `listener_too` a listener param is automated per instance as well (then
    the per-frame geometry kernel consumes the device replay of the listener and the value blocks of the panner)."""
    sr, n, frames = 48000.0, 4, RQ * 40 + 3
    noise = white_noise(n, 1, frames, seed0=31)
    outs = []
    for b in (hip, orc):
        ctx = waa.OfflineAudioContext(2, frames, sr, n_instances=n, binding=b)
        src = ctx.create_buffer_source()
        src.set_buffer_batch(noise, sr)
        pn = ctx.create_panner(distance_model="inverse", rolloff_factor=1.2, position=(0.5, 0.2, -1.0))
        for i in range(n):
            pn.position_x.set_value_at_time(-2.0 + i, 0.0, instance=i)
            pn.position_x.linear_ramp_to_value_at_time(3.0 - 0.5 * i, frames / sr * (0.5 + 0.1 * i), instance=i)
            if listener_too:
                ctx.listener().position_z.set_value_at_time(0.5 * i, RQ * 5 / sr, instance=i)
                ctx.listener().position_z.linear_ramp_to_value_at_time(-1.0 - i, RQ * 30 / sr, instance=i)
        src.connect(pn).connect(ctx.destination())
        src.start()
        outs.append(ctx.start_rendering_sync().data)
        ctx.close()
    assert rms_err(*outs).max() <= TOL
    assert np.abs(outs[0] - outs[1]).max() <= 5e-6
