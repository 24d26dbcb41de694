"""EVERY instance of the full-size BASELINE batches against the oracle (-m gpu).

The sampled full-size tests of test_gpu_parity.py pick the first, last and middle contexts; the defects the round-3 fuzzing
found were exactly of the instance-dependent kind those picks can miss (crosstalk between the two contexts of one packed
convolver transform, a partial last tile, reader counting through aliases).  Here the device renders the whole batch once and
the oracle — one context per host thread on every CPU this process may use (orc_set_threads) — renders ALL contexts, in
chunks that bound host memory; every (instance, channel) row is compared: SURVEY.md section 8(d) "for every instance",
1e-6 RMS per channel (the north star's tolerance) plus a max-abs bound per workload.

Cost on the GPU box (16 usable CPUs): C2 / C5 seconds each, C3 / C4 ~10 s each, T1 ~20-25 s of oracle time."""
import numpy as np
import pytest

import web_audio_api_rs_amd as waa
from every_instance import _compare_all, _oracle_chunks
from graphs import assert_all_finite, assert_le, c2, c4, c5, garage_ir, rms_err, strict_max, t1

pytestmark = pytest.mark.gpu
TOL = 1e-6
FRAMES = 480000  # 10 s at 48 kHz = 3750 render quanta


def _noise(n_inst, n_ch, frames, seed):
    rng = np.random.default_rng(seed)
    out = rng.random((n_inst, n_ch, frames), dtype=np.float32)
    out *= 2.0
    out -= 1.0
    return out


def test_c2_every_instance(hip, orc):
    """BASELINE config 2 (the bench's headline batch): 1024 contexts x 10 s, BufferSource -> Biquad(lowpass 200 Hz, Q 1) ->
    Gain(0.5) -> destination; all 1024 contexts against the oracle."""
    n_inst = 1024
    noise = _noise(n_inst, 2, FRAMES, 123)
    ctx, _ = c2(hip, noise)
    out = ctx.start_rendering_sync().data
    ctx.close()
    assert out.shape == (n_inst, 2, FRAMES)
    _compare_all(orc, out, lambda be, lo, hi: c2(be, noise[lo:hi]), chunk=256, max_abs=1e-7)


def test_c2_every_instance_with_its_own_filter_and_gain(hip, orc):
    """the same batch with per-instance AudioParam values (cutoff 40 Hz .. 18 kHz, Q, gain): the per-instance coefficient
    and gain tables are indexed by every one of the 1024 contexts"""
    n_inst = 1024
    noise = _noise(n_inst, 2, FRAMES, 124)
    rng = np.random.default_rng(5)
    f = np.geomspace(40.0, 18000.0, n_inst).astype(np.float32)
    q = rng.uniform(0.3, 8.0, n_inst).astype(np.float32)
    g = rng.uniform(0.1, 1.5, n_inst).astype(np.float32)

    def build(be, lo, hi):
        ctx, nodes = c2(be, noise[lo:hi])
        for i in range(lo, hi):
            nodes["biquad"].frequency.set_value(float(f[i]), instance=i - lo)
            nodes["biquad"].q.set_value(float(q[i]), instance=i - lo)
            nodes["gain"].gain.set_value(float(g[i]), instance=i - lo)
        return ctx, nodes

    ctx, _ = build(hip, 0, n_inst)
    out = ctx.start_rendering_sync().data
    ctx.close()
    _compare_all(orc, out, build, chunk=256, max_abs=2e-7)


@pytest.mark.parametrize("rate,buf_sr,n_ch", [(1.5, None, 2), (1.0, 38000.0, 2), (1.5, None, 1)])
def test_c5_every_instance(hip, orc, rate, buf_sr, n_ch):
    """BASELINE config 5: 2048 contexts x 10 s, looping BufferSource (playbackRate 1.5 / a 38 kHz buffer in a 48 kHz context)
    -> WaveShaper(2048-point curve) -> destination; SURVEY.md section 8(d) C5: 1e-6 RMS per channel FOR EVERY INSTANCE."""
    n_inst, buf_frames = 2048, 65536
    noise = _noise(n_inst, n_ch, buf_frames, 33)
    ctx, _ = c5(hip, noise, length=FRAMES, rate=rate, buf_sr=buf_sr, loop=True)
    out = ctx.start_rendering_sync().data
    ctx.close()
    assert out.shape == (n_inst, 2, FRAMES)
    _compare_all(orc, out, lambda be, lo, hi: c5(be, noise[lo:hi], length=FRAMES, rate=rate, buf_sr=buf_sr, loop=True),
                 chunk=256, max_abs=1e-6)


def test_c3_every_instance_real_ir(hip, orc):
    """BASELINE config 3: 512 contexts x 10 s, BufferSource -> Convolver(parking-garage IR, normalised) -> destination;
    all 512 contexts (256 packed pairs) against the restated fft-convolver."""
    n_inst = 512
    noise = _noise(n_inst, 2, FRAMES, 31)
    ctx, _ = t1(hip, noise, garage_ir(hip), with_biquad=False)
    assert "P=22 blocks=59" in ctx.plan_describe()
    out = ctx.start_rendering_sync().data
    ctx.close()
    ir = garage_ir(orc)
    _compare_all(orc, out, lambda be, lo, hi: t1(be, noise[lo:hi], ir, with_biquad=False), chunk=128, max_abs=2e-6)


def test_t1_every_instance_real_ir(hip, orc):
    """The north-star graph at its size: 1024 contexts x 10 s, BufferSource -> Biquad -> Convolver(real IR) -> destination
    (bench.py's `t1`); all 1024 contexts = both members of all 512 packed pairs."""
    n_inst = 1024
    noise = _noise(n_inst, 2, FRAMES, 34)
    ctx, _ = t1(hip, noise, garage_ir(hip))
    plan = ctx.plan_describe()
    assert "P=22 blocks=59 pairs=512" in plan and "the Biquad in front, in the impulse response" in plan
    out = ctx.start_rendering_sync().data
    ctx.close()
    ir = garage_ir(orc)
    _compare_all(orc, out, lambda be, lo, hi: t1(be, noise[lo:hi], ir), chunk=128, max_abs=2e-6)


def test_t1_every_instance_with_its_own_filter(hip, orc):
    """T1 with a cutoff per context (40 Hz .. 12 kHz): per-context coefficients cannot live in the shared impulse response, the
    Biquad stays the exact-order filter stage of the forward transform (conv_fft3_fwd_bq_kernel, round 3's form) — every one of
    the 1024 contexts against the oracle."""
    n_inst = 1024
    noise = _noise(n_inst, 2, FRAMES, 36)
    f = np.geomspace(40.0, 12000.0, n_inst).astype(np.float32)

    def build(be, lo, hi):
        ctx, nodes = t1(be, noise[lo:hi], garage_ir(be))
        for i in range(lo, hi):
            nodes["biquad"].frequency.set_value(float(f[i]), instance=i - lo)
        return ctx, nodes

    ctx, _ = build(hip, 0, n_inst)
    plan = ctx.plan_describe()
    assert "P=22 blocks=59 pairs=512" in plan and "the Biquad in front, in the forward transform" in plan
    out = ctx.start_rendering_sync().data
    ctx.close()
    _compare_all(orc, out, build, chunk=128, max_abs=2e-6)


def test_c4_every_instance_real_ir_and_every_analyser_pull(hip, orc):
    """BASELINE config 4, one GPU's shard: 512 contexts x 10 s, BufferSource -> Biquad -> Convolver -> StereoPanner(0.1) ->
    Analyser(2048, 0.8) -> destination; every context's render AND every context's analyser pull (time-domain data and
    spectrum, one batched pull on the device) against the oracle's."""
    n_inst = 512
    noise = _noise(n_inst, 2, FRAMES, 32)
    ctx, nodes = c4(hip, noise, garage_ir(hip))
    out = ctx.start_rendering_sync().data
    gf = nodes["analyser"].get_float_frequency_data_all()
    gt = nodes["analyser"].get_float_time_domain_data_all()
    ctx.close()
    ir = garage_ir(orc)
    assert_all_finite(out, "device render")
    assert_all_finite(gt, "device time-domain pulls")
    # spectrum pulls are dB values: -inf (an exactly zero magnitude) is a legal value, NaN and +inf are not
    assert not np.isnan(gf).any() and not np.isposinf(gf).any(), "device spectrum pull holds NaN / +inf"
    worst_rms, worst_t, worst_lin, worst_db = 0.0, 0.0, 0.0, 0.0
    for lo, hi, octx, onodes in _oracle_chunks(orc, lambda be, lo, hi: c4(be, noise[lo:hi], ir), n_inst, 128):
        ref = octx.start_rendering_sync().data
        of = np.stack([onodes["analyser"].get_float_frequency_data(instance=i) for i in range(hi - lo)])
        ot = np.stack([onodes["analyser"].get_float_time_domain_data(instance=i) for i in range(hi - lo)])
        octx.close()
        assert_all_finite(ref, "oracle render")
        assert_all_finite(ot, "oracle time-domain pulls")
        assert not np.isnan(of).any() and not np.isposinf(of).any()
        worst_rms = strict_max(worst_rms, rms_err(out[lo:hi], ref).max())
        worst_t = strict_max(worst_t, np.abs(gt[lo:hi] - ot).max())
        # linear magnitudes: 10^(-inf / 20) = 0, so a -inf bin on either side is compared as the zero it stands for
        gl, ol = 10.0 ** (gf[lo:hi].astype(np.float64) / 20), 10.0 ** (of.astype(np.float64) / 20)
        row_peak = ol.max(axis=1)
        assert (row_peak > 0).all()
        worst_lin = strict_max(worst_lin, (np.abs(gl - ol).max(axis=1) / row_peak).max())
        # dB values (what the getter returns): a linear error e relative to the row's peak is 20 log10(1 + e 10^(D/20)) dB on
        # a bin D dB below the peak — compared on the bins within 60 dB of the peak, where the linear tolerance allows 0.017 dB
        # (farther down the f32 transform's own roundoff IS the value, analysis.rs:301-345 and the device alike: 0.08 dB at
        # -100 dB measured, 1e-2 relative at -100 dB = 1e-7 of the peak).  The oracle's loud bins are finite by construction;
        # a device -inf there gives an infinite difference and fails.  (Indexing BEFORE subtracting: -inf - -inf on a far bin
        # was the RuntimeWarning of round 4's log.)
        loud = of > (of.max(axis=1, keepdims=True) - 60.0)
        worst_db = strict_max(worst_db, np.abs(gf[lo:hi][loud].astype(np.float64) - of[loud]).max())
    print(f"C4 all {n_inst}: render RMS {worst_rms:.3e}, time-domain pull max |diff| {worst_t:.3e}, "
          f"spectrum: linear diff / row peak {worst_lin:.3e}, dB diff on bins within 60 dB of the peak {worst_db:.3e}")
    assert_le(worst_rms, TOL)
    assert_le(worst_t, 2e-6)
    assert_le(worst_lin, ANALYSER_LIN_TOL)
    assert_le(worst_db, ANALYSER_DB_TOL)


# Analyser tolerances (analysis.rs:278-345: Blackman window, f32 real FFT of fft_size points, magnitude / fft_size,
# smoothing, 20 log10).  Both sides run an f32 FFT of 2048 points over the same time-domain data (which itself differs by
# <= 7e-8): the rounding of an f32 transform is ~ sqrt(log2 N) * 6e-8 ~ 2e-7 relative to the row's PEAK magnitude per side,
# two different factorizations (the oracle's radix-2, the device's packed real transform) and the smoothing recursion over
# the pulls add up to the 1.8e-6 measured over all 512 contexts (profiles/r04a_allinst.log); 4e-6 is asserted.  The dB bound
# follows from it: 20 log10(1 + 4e-6 * 10^(60/20)) = 0.035 dB on the bins within 60 dB of the peak.
ANALYSER_LIN_TOL = 4e-6
ANALYSER_DB_TOL = 0.035


@pytest.mark.parametrize("oversample,n_inst", [("2x", 1024), ("4x", 512)])
def test_oversampled_waveshaper_every_instance(hip, orc, oversample, n_inst):
    """bench.py's os2 / os4 batches (BufferSource(stereo) -> WaveShaper(tanh 2049-pt, oversample) -> destination, 10 s): the
    transform form of round 4 (waa_osfft.hip) walks every instance in runs of ~235 quanta whose overlaps are recomputed at
    the run heads — every context and every run boundary against the oracle's stage-by-stage f32 FFT resamplers.  Sources
    start at per-instance times (skipped quanta in front) and a third of them stop before the render ends."""
    noise = _noise(n_inst, 2, FRAMES, 35)
    curve = np.tanh(np.linspace(-3.0, 3.0, 2049)).astype(np.float32)

    def build(be, lo, hi):
        ctx = waa.OfflineAudioContext(2, FRAMES, 48000.0, n_instances=hi - lo, binding=be)
        src = ctx.create_buffer_source()
        src.set_buffer_batch(noise[lo:hi], 48000.0)
        src.connect(ctx.create_wave_shaper(curve=curve, oversample=oversample)).connect(ctx.destination())
        for i in range(lo, hi):
            src.start_at((i % 11) * 0.013, instance=i - lo)
            if i % 3 == 0:
                src.stop_at(6.0 + (i % 7) * 0.37, instance=i - lo)
        return ctx, {}

    ctx, _ = build(hip, 0, n_inst)
    assert "256-point transforms per quantum in one launch" in ctx.plan_describe()
    out = ctx.start_rendering_sync().data
    ctx.close()
    _compare_all(orc, out, build, chunk=128, max_abs=4e-6)
