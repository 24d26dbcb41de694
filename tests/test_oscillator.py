"""OscillatorNode: the reference's tests re-typed (src/node/oscillator.rs:700-1460,
src/periodic_wave.rs:213-320) on both backends plus GPU-vs-oracle parity.  Note: the reference disables polyBLEP
under cfg!(test) (oscillator.rs:626-632); the restatement is the PRODUCTION code (polyBLEP on), so the crude
square/sawtooth comparisons of the reference's test build are replaced by the polyBLEP formula itself."""
import numpy as np
import pytest

import web_audio_api_rs_amd as waa
from graphs import rms_err, white_noise

RQ = 128


def render_osc(be, sr, length, freq, type_="sine", start=0.0, stop=None, detune=0.0, wave=None):
    c = waa.OfflineAudioContext(1, length, float(sr), binding=be)
    kw = dict(type_=type_, frequency=freq, detune=detune)
    if wave is not None:
        kw = dict(frequency=freq, detune=detune, periodic_wave=wave)
    osc = c.create_oscillator(**kw)
    osc.connect(c.destination())
    osc.start_at(start)
    if stop is not None:
        osc.stop_at(stop)
    return c.start_rendering_sync().data[0, 0]


def ref_phase_sine(n, freq, sr, phase0=0.0, first=0):
    out = np.zeros(n, np.float32)
    phase, incr = phase0, float(np.float32(freq)) / float(sr)
    for i in range(first, n):
        out[i] = np.float32(np.sin(phase * 2.0 * np.pi))
        phase += incr
        if phase >= 1.0:
            phase -= 1.0
    return out


@pytest.mark.parametrize("exp", range(5))
def test_sine_raw(be, exp):
    """oscillator.rs:806-840: 1 .. 10 kHz for one second, abs_all <= 1e-5 (8192 frames here)"""
    freq, sr, n = 10.0 ** exp, 44100, 8192
    out = render_osc(be, sr, n, freq)
    assert np.max(np.abs(out - ref_phase_sine(n, freq, sr))) <= 1e-5


def test_sine_negative_frequency(be):
    """oscillator.rs:1430-1455"""
    sr, n, freq = 44100, 4096, -440.0
    out = render_osc(be, sr, n, freq)
    i = np.arange(n)
    exp = np.sin(2 * np.pi * freq * i / sr).astype(np.float32)
    assert np.max(np.abs(out - exp)) <= 1e-4


def test_triangle_raw(be):
    """oscillator.rs:909-953: abs_all <= 1e-10"""
    sr, n, freq = 44100, 4096, 100.0
    out = render_osc(be, sr, n, freq, type_="triangle")
    exp = np.zeros(n, np.float32)
    phase, incr = 0.0, float(np.float32(freq)) / sr
    for i in range(n):
        s = -4.0 * phase + 2.0
        if s > 1.0:
            s = 2.0 - s
        elif s < -1.0:
            s = -2.0 - s
        exp[i] = np.float32(s)
        phase += incr
        if phase >= 1.0:
            phase -= 1.0
    assert np.max(np.abs(out - exp)) <= 1e-10


def poly_blep(t, dt):
    """oscillator.rs:626-643"""
    if t < dt:
        t /= dt
        return t + t - t * t - 1.0
    if t > 1.0 - dt:
        t = (t - 1.0) / dt
        return t * t + t + t + 1.0
    return 0.0


@pytest.mark.parametrize("type_", ["square", "sawtooth"])
def test_square_and_sawtooth_with_polyblep(be, type_):
    """oscillator.rs:588-602 (production arithmetic; the polyBLEP term itself is pinned by :1095-1132)"""
    sr, n, freq = 44100, 4096, 441.0
    out = render_osc(be, sr, n, freq, type_=type_)
    exp = np.zeros(n, np.float32)
    phase, incr = 0.0, float(np.float32(freq)) / sr
    unroll = lambda p: p - 1.0 if p >= 1.0 else (p + 1.0 if p < 0.0 else p)
    for i in range(n):
        if type_ == "square":
            s = (1.0 if phase < 0.5 else -1.0) + poly_blep(phase, incr) - poly_blep(unroll(phase + 0.5), incr)
        else:
            ph = unroll(phase + 0.5)
            s = 2.0 * ph - 1.0 - poly_blep(ph, incr)
        exp[i] = np.float32(s)
        phase = unroll(phase + incr)
    assert np.max(np.abs(out - exp)) <= 1e-6


def test_polyblep_isolated(orc_lib):
    """oscillator.rs:1095-1132 is a unit test of poly_blep; covered through the rendered waveforms above"""
    assert poly_blep(0.0, 0.01) == -1.0 and poly_blep(0.5, 0.01) == 0.0


@pytest.mark.parametrize("harmonics", [1, 2])
def test_periodic_wave(be, harmonics):
    """oscillator.rs:1001-1092: custom wave = sine (+ 0.5 second harmonic), normalised; abs_all <= 1e-5"""
    sr, n, freq = 44100, 4096, 100.0
    real = [0.0] * (harmonics + 1)
    imag = [0.0, 1.0] + ([0.5] if harmonics == 2 else [])
    out = render_osc(be, sr, n, freq, wave=waa.PeriodicWave(real=real, imag=imag))
    i = np.arange(8192, dtype=np.float64)
    table = np.sin(2 * np.pi * i / 8192) + (0.5 * np.sin(4 * np.pi * i / 8192) if harmonics == 2 else 0.0)
    norm = 1.0 / np.abs(table).max()
    phase, incr = 0.0, float(np.float32(freq)) / sr
    exp = np.zeros(n, np.float32)
    for k in range(n):
        exp[k] = np.float32(norm * (np.sin(phase * 2 * np.pi) + (0.5 * np.sin(phase * 4 * np.pi) if harmonics == 2 else 0.0)))
        phase += incr
        if phase >= 1.0:
            phase -= 1.0
    assert np.max(np.abs(out - exp)) <= 1e-4  # 8192-point table, linear interpolation (reference: 1e-5 at lower freq)


def test_finished_wavetable_equals_the_coefficient_form(be):
    """waa_oscillator_set_wavetable (what the Rust shim forwards: the reference's processor only holds the finished table,
    oscillator.rs:487-493): the table periodic_wave.rs:174-196 generates in f32, handed over as it is, renders bit for bit what
    the coefficient form renders when the host's cos / sin agree with the library's — they need not to the last ulp, so the
    bound here is the effect of one ulp of table error; the exact statement is the second half: a table is USED as given
    (linear interpolation `prev.mul_add(1 - k, next * k)`, oscillator.rs:623-636)."""
    sr, n, freq = 44100, 4096, 317.0
    real, imag = [0.0, 0.3, 0.0, 0.1], [0.0, 1.0, 0.5, 0.0]
    a = render_osc(be, sr, n, freq, wave=waa.PeriodicWave(real=real, imag=imag))
    i = np.arange(8192, dtype=np.float32)
    phase = np.float32(2.0 * np.float32(np.pi)) * i / np.float32(8192)
    table = np.zeros(8192, np.float32)
    for j in range(1, 4):
        rad = phase * np.float32(j)
        table += np.float32(real[j]) * np.cos(rad) + np.float32(imag[j]) * np.sin(rad)
    table *= np.float32(1.0) / np.abs(table).max()
    b = render_osc(be, sr, n, freq, wave=waa.PeriodicWave.from_wavetable(table))
    assert np.max(np.abs(a - b)) <= 2e-6
    # a random table: the lookup itself, restated
    rng = np.random.default_rng(4)
    table = rng.uniform(-1, 1, 8192).astype(np.float32)
    out = render_osc(be, sr, n, freq, wave=waa.PeriodicWave.from_wavetable(table))
    exp = np.zeros(n, np.float32)
    ph, incr = 0.0, float(np.float32(freq)) / sr
    for k in range(n):
        pos = ph * 8192
        lo = int(pos)
        kk = np.float32(pos - lo)
        exp[k] = np.float32(np.float64(table[lo % 8192]) * np.float64(np.float32(1.0) - kk) + np.float64(np.float32(table[(lo + 1) % 8192] * kk)))
        ph += incr
        if ph >= 1.0:
            ph -= 1.0
    assert np.max(np.abs(out - exp)) <= 2e-7
    with pytest.raises(waa.WaaError, match="IndexSizeError"):
        waa.PeriodicWave.from_wavetable(np.zeros(100, np.float32))


def test_periodic_wave_validation(be):
    """periodic_wave.rs:104-139"""
    with pytest.raises(waa.WaaError, match="IndexSizeError"):
        waa.PeriodicWave(real=[0.0], imag=[0.0])
    with pytest.raises(waa.WaaError, match="IndexSizeError"):
        waa.PeriodicWave(real=[0.0, 1.0], imag=[0.0, 1.0, 0.5])
    c = waa.OfflineAudioContext(1, RQ, 44100.0, binding=be)
    osc = c.create_oscillator()
    with pytest.raises(waa.WaaError, match="InvalidStateError"):
        osc.set_type("custom")  # oscillator.rs:305-309
    osc.set_periodic_wave(waa.PeriodicWave())
    osc.set_type("square")      # ignored, oscillator.rs:770-797
    assert osc.type_ == "custom"


def test_sub_quantum_and_sub_sample_start(be):
    """oscillator.rs:1135-1197"""
    sr = 44100
    out = render_osc(be, sr, 4096, 1.25, start=2.0 / sr)
    assert np.max(np.abs(out - ref_phase_sine(4096, 1.25, sr, first=2))) <= 1e-5
    sr = 96000
    out = render_osc(be, sr, 4096, 1.0, start=1.3 / sr)
    incr = 1.0 / sr
    assert out[0] == 0.0 and out[1] == 0.0
    assert np.max(np.abs(out - ref_phase_sine(4096, 1.0, sr, phase0=0.7 * incr, first=2))) <= 1e-5


def test_sub_quantum_stop_and_disarm(be):
    """oscillator.rs:1199-1246"""
    sr = 44100
    out = render_osc(be, sr, 1024, 2345.6, stop=6.0 / sr)
    exp = ref_phase_sine(1024, 2345.6, sr)
    exp[6:] = 0.0
    assert np.max(np.abs(out - exp)) <= 1e-5
    out = render_osc(be, sr, 128, 440.0, start=1.0, stop=0.5)  # stop before start: silence
    assert np.array_equal(out, np.zeros(128, np.float32))


def test_start_in_the_past_is_now(be):
    """oscillator.rs:1310-1342 (start_at(0) is the same thing offline); delayed start :1409-1428"""
    sr = 48000
    out = render_osc(be, sr, RQ * 3, 440.0, start=RQ / sr)
    exp = np.zeros(RQ * 3, np.float32)
    exp[RQ:] = ref_phase_sine(RQ * 2, 440.0, sr)
    assert np.max(np.abs(out - exp)) <= 1e-5


@pytest.mark.parametrize("freq", [30000.0, -30000.0])
def test_outside_nyquist_is_silent(be, freq):
    """oscillator.rs:1344-1382: the param clamps to +-nyquist, at which the oscillator outputs zero"""
    out = render_osc(be, 44100, 128, freq)
    assert np.max(np.abs(out)) <= 1e-5


# --------------------------------------------------------------------------- GPU parity
def _fm_patch(binding, n, frames):
    """two-operator FM: modulator oscillator -> gain (index) -> carrier.frequency; detuned per instance; envelope"""
    c = waa.OfflineAudioContext(2, frames, 48000.0, n_instances=n, binding=binding)
    mod = c.create_oscillator(type_="sine", frequency=110.0)
    idx = c.create_gain(gain=300.0)
    car = c.create_oscillator(type_="sine", frequency=440.0)
    sq = c.create_oscillator(type_="square", frequency=55.0)
    saw = c.create_oscillator(type_="sawtooth", frequency=82.4)
    tri = c.create_oscillator(type_="triangle", frequency=220.0, detune=700.0)
    for i in range(n):
        car.detune.set_value(25.0 * i, instance=i)
        mod.frequency.set_value(110.0 + 7.0 * i, instance=i)
    mix = c.create_gain(gain=0.2)
    pan = c.create_stereo_panner(pan=0.25)
    mod.connect(idx).connect(car.frequency)
    for o in (car, sq, saw, tri):
        o.connect(mix)
    mix.connect(pan).connect(c.destination())
    mod.start()
    car.start_at(0.001)
    sq.start()
    saw.start_at(300.5 / 48000.0)
    saw.stop_at(0.05)
    tri.start()
    out = c.start_rendering_sync().data
    c.close()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("exact", [False, True])
def test_parity_fm_patch(hip, orc, exact, monkeypatch):
    """unmodulated oscillators use the time-parallel kernel (closed-form phase), the FM carrier the prefix-sum
    kernel; WAA_OSC_EXACT forces the serial lane-per-instance kernel for both"""
    if exact:
        monkeypatch.setenv("WAA_OSC_EXACT", "1")
    n, frames = 5, 2048 * 2 + 99
    g, o = _fm_patch(hip, n, frames), _fm_patch(orc, n, frames)
    assert rms_err(g, o).max() <= 1e-6
    assert np.abs(g - o).max() <= 5e-6


@pytest.mark.gpu
def test_parity_long_render_closed_form_phase(hip, orc):
    """10 s: the closed-form phase of the time-parallel kernel against 480 000 rounded additions of the reference;
    k-rate frequency sweep, start and stop inside quanta"""
    sr, frames = 48000.0, 480000
    nq = (frames + RQ - 1) // RQ
    outs = []
    for be_ in (hip, orc):
        c = waa.OfflineAudioContext(1, frames, sr, n_instances=3, binding=be_)
        o1 = c.create_oscillator(type_="sawtooth", frequency=97.3)
        o2 = c.create_oscillator(type_="sine", frequency=440.0)
        o2.frequency.set_block(0, np.geomspace(100.0, 6000.0, nq).astype(np.float32))
        o3 = c.create_oscillator(type_="square", frequency=-333.3, detune=50.0)
        mix = c.create_gain(gain=0.3)
        for o_ in (o1, o2, o3):
            o_.connect(mix)
        mix.connect(c.destination())
        o1.start_at(0.01234)
        o1.stop_at(9.4321)
        o2.start()
        o3.start_at(1.0)
        outs.append(c.start_rendering_sync().data)
        c.close()
    assert rms_err(*outs).max() <= 1e-6
    assert np.abs(outs[0] - outs[1]).max() <= 2e-5  # isolated samples on a square edge may land on the other side


@pytest.mark.gpu
def test_parity_fm_long_render(hip, orc):
    """4 s of two-operator FM with a sub-sample start and a stop: prefix-sum phase vs the reference's running sum"""
    sr, frames = 48000.0, 48000 * 4
    outs = []
    for be_ in (hip, orc):
        c = waa.OfflineAudioContext(1, frames, sr, n_instances=2, binding=be_)
        mod = c.create_oscillator(type_="sine", frequency=3.0)
        idx = c.create_gain(gain=200.0)
        car = c.create_oscillator(type_="sawtooth", frequency=220.0)
        car.detune.set_value(300.0, instance=1)
        mod.connect(idx).connect(car.frequency)
        car.connect(c.destination())
        mod.start()
        car.start_at(0.2500071)
        car.stop_at(3.5)
        outs.append(c.start_rendering_sync().data)
        c.close()
    assert np.abs(outs[1]).max() > 0.5
    assert rms_err(*outs).max() <= 1e-6
    assert np.abs(outs[0] - outs[1]).max() <= 1e-4  # isolated samples next to a sawtooth edge


# --------------------------------------------------------------------------- more of the reference's tests (oracle pins)
@pytest.mark.parametrize("exp", range(5))
def test_sine_raw_exact_phase(orc, exp):
    """oscillator.rs:842-869: the expected phase is freq * i / sr computed per sample (no accumulation), one second,
    abs_all <= 1e-5 — bounds the drift of the running phase"""
    sr = 44100
    freq = float(np.float32(10.0) ** np.float32(exp))
    out = render_osc(orc, sr, sr, freq)
    i = np.arange(sr, dtype=np.float64)
    want = np.sin(freq * i / sr * 2.0 * np.pi).astype(np.float32)
    assert np.max(np.abs(out - want)) <= 1e-5


def test_sub_sample_stop(orc):
    """oscillator.rs:1278-1308: stop_at(19.4 / sr): frames 0..19 sound, the rest is silent"""
    sr = 44100
    out = render_osc(orc, sr, 2048, 8910.1, stop=19.4 / sr)
    want = ref_phase_sine(2048, 8910.1, sr)
    want[20:] = 0.0
    assert np.max(np.abs(out - want)) <= 1e-5


def test_reenters_the_audible_range_after_large_phase_increments(orc):
    """oscillator.rs:1384-1407: 20 kHz detuned by +2400 cents is 80 kHz (>= nyquist: silence); the detune drops to 0 at
    the second quantum and the oscillator must come back with finite, non-zero samples"""
    sr = 44100
    c = waa.OfflineAudioContext(1, 256, float(sr), binding=orc)
    osc = c.create_oscillator(frequency=20000.0, detune=2400.0)
    osc.detune.set_value_at_time(0.0, RQ / sr)
    osc.connect(c.destination())
    osc.start_at(0.0)
    out = c.start_rendering_sync().data[0, 0]
    assert np.max(np.abs(out[:RQ])) <= 1e-5
    assert np.all(np.isfinite(out[RQ:])) and np.any(out[RQ:] != 0.0)


@pytest.mark.measure
@pytest.mark.gpu
@pytest.mark.parametrize("modulated", [False, True])
def test_folded_post_ops_are_bit_identical(hip, orc, modulated, monkeypatch):
    """Oscillator -> Gain -> Gain -> stereo destination: the gains and the up-mix 1 -> 2 are rendered by the oscillator's own
    launch (time-parallel and prefix-sum kernels); same bits as the chain launch behind it, same samples as the oracle"""
    def render(be_):
        c = waa.OfflineAudioContext(2, RQ * 37 + 11, 48000.0, n_instances=3, binding=be_)
        osc = c.create_oscillator(type_="sawtooth", frequency=220.0)
        if modulated:
            lfo = c.create_oscillator(frequency=6.0)
            lfo.connect(c.create_gain(gain=40.0)).connect(osc.frequency)
            lfo.start()
        g1 = c.create_gain(gain=0.5)
        for i in range(3):
            g1.gain.set_value(0.25 * (i + 1), instance=i)
        osc.connect(g1).connect(c.create_gain(gain=0.8)).connect(c.destination())
        osc.start_at(0.0013)
        osc.stop_at(0.09)
        plan = c.plan_describe() if be_ is hip else ""
        out = c.start_rendering_sync().data
        c.close()
        return out, plan

    fused, plan = render(hip)
    assert "renders 2 gain(s) and the up-mix 1 -> 2" in plan
    monkeypatch.setenv("WAA_NO_OSC_POST", "1")
    plain, plan = render(hip)
    assert "renders 2 gain(s)" not in plan
    ref, _ = render(orc)
    assert np.array_equal(fused, plain)
    assert np.array_equal(fused[:, 0], fused[:, 1]) and np.any(fused != 0)
    err = np.sqrt(np.mean((fused.astype(np.float64) - ref) ** 2, axis=2))
    assert err.max() <= 1e-6


def _osc_behind_an_analyser(be, device=-1):
    """Oscillator -> Analyser -> two Gains -> destination: the analyser aliases the oscillator's signal, both Gains read it"""
    kw = {"device": device} if device != -1 else {}
    c = waa.OfflineAudioContext(2, 128 * 40, 48000.0, n_instances=3, binding=be, **kw)
    osc = c.create_oscillator(type_="sawtooth", frequency=330.0)
    an = c.create_analyser(fft_size=256)
    a, b = c.create_gain(gain=0.5), c.create_gain(gain=0.25)
    for i in range(3):
        a.gain.set_value(0.2 + 0.3 * i, instance=i)
    osc.connect(an)
    an.connect(a).connect(c.destination())
    an.connect(b).connect(c.destination())
    osc.start()
    return c, an


def test_plan_oscillator_keeps_its_signal_when_an_alias_has_other_readers(hip):
    """the oscillator may render a Gain chain itself only if NOTHING else reads its signal — also not through a node that
    aliases it (fuzz seed 502310 of the frozen-state generator, found with WAA_POISON_ALLOC)"""
    c, _ = _osc_behind_an_analyser(hip, device=waa.PLAN_ONLY)
    plan = c.plan_describe()
    assert "chain itself" not in plan, plan
    c.close()


@pytest.mark.gpu
def test_parity_oscillator_behind_an_analyser_with_two_readers(hip, orc):
    outs, bins = [], []
    for be in (hip, orc):
        c, an = _osc_behind_an_analyser(be)
        outs.append(c.start_rendering_sync().data)
        bins.append(an.get_float_frequency_data(instance=1))
        c.close()
    assert np.abs(outs[0] - outs[1]).max() <= 2e-5 and np.abs(outs[1]).max() > 0.1
    finite = np.isfinite(bins[1])
    assert np.abs(bins[0][finite] - bins[1][finite]).max() <= 0.05


@pytest.mark.measure
def test_plan_replay_shortcut_agrees_with_the_frame_walk(hip, monkeypatch):
    """plan_oscillator skips the 128-frame walk of a quantum that lies wholly inside [start, stop) (it was 2.4 s of a
    1024-context plan); with WAA_OSC_PLAN_CHECK=1 every quantum is walked anyway and a shortcut that would have answered
    differently fails the plan (status 3).  Random start / stop times — on quantum boundaries, a frame or a fraction of a
    frame beside them, inside one quantum, never — for the constant-frequency (closed-form) and the a-rate (prefix-sum) plans."""
    monkeypatch.setenv("WAA_OSC_PLAN_CHECK", "1")
    rng = np.random.default_rng(77)
    sr = 48000.0
    for trial in range(60):
        frames = int(rng.integers(1, 40)) * 128 + int(rng.integers(0, 128))
        n = 8
        c = waa.OfflineAudioContext(1, frames, sr, n_instances=n, binding=hip, device=waa.PLAN_ONLY)
        osc = c.create_oscillator(type_="sawtooth", frequency=float(rng.uniform(20.0, 9000.0)))
        if trial % 2:
            osc.frequency.set_value_at_time(100.0, 0.0)
            osc.frequency.linear_ramp_to_value_at_time(3000.0, frames / sr)
        for i in range(n):
            q0 = int(rng.integers(0, max(1, frames // 128)))
            kind = int(rng.integers(0, 6))
            start = [0.0, q0 * 128 / sr, (q0 * 128 + 1) / sr, (q0 * 128 - 0.25) / sr, (q0 * 128 + 64.5) / sr,
                     float(np.nextafter(q0 * 128 / sr, 1.0))][kind]
            osc.start_at(max(start, 0.0), instance=i)
            if rng.random() < 0.7:
                q1 = q0 + int(rng.integers(0, 12))
                kind = int(rng.integers(0, 6))
                stop = [(q1 + 1) * 128 / sr, ((q1 + 1) * 128 + 1) / sr, ((q1 + 1) * 128 - 1) / sr, (q1 * 128 + 77.3) / sr,
                        float(np.nextafter((q1 + 1) * 128 / sr, 0.0)), float(np.nextafter((q1 + 1) * 128 / sr, 1.0))][kind]
                osc.stop_at(max(stop, max(start, 0.0)), instance=i)
        osc.connect(c.destination())
        assert "oscillator node" in c.plan_describe()
        c.close()


def _two_operator(binding, n, frames, variant, plan_only=False):
    """modulator -> [Gain (index)] -> carrier.frequency -> destination, in the variants the FM fold has to tell apart:
    plain        sine modulator, constant index (folded into the carrier's kernel)
    square-mod   a band-limited square as the modulator, per-instance index, modulator starting late and stopping early
    ramped-index the index and the carrier's intrinsic frequency ramp (a-rate params: values per frame — not folded)
    no-gain      the modulator connected straight to the param
    clamped      an index large enough to drive the sum past the param's range (the clamp of mix_to_output) and below zero
    shared       the modulator ALSO reaches the destination (it has a second reader: not folded)
    two-mods     two modulators summed on the param (more than one input: not folded)"""
    kw = {"device": waa.PLAN_ONLY} if plan_only else {}
    c = waa.OfflineAudioContext(1, frames, 48000.0, n_instances=n, binding=binding, **kw)
    mod = c.create_oscillator(type_="square" if variant == "square-mod" else "sine", frequency=110.0)
    car = c.create_oscillator(type_="sine", frequency=440.0)
    for i in range(n):
        mod.frequency.set_value(90.0 + 13.0 * i, instance=i)
        car.detune.set_value(30.0 * i, instance=i)
    head = mod
    if variant != "no-gain":
        idx = c.create_gain(gain=300.0)
        if variant == "square-mod":
            for i in range(n):
                idx.gain.set_value(100.0 + 150.0 * i, instance=i)
        if variant == "ramped-index":
            idx.gain.set_value_at_time(50.0, 0.0).linear_ramp_to_value_at_time(900.0, frames / 48000.0)
            car.frequency.set_value_at_time(440.0, 0.0).linear_ramp_to_value_at_time(880.0, frames / 48000.0 * 0.7)
        if variant == "clamped":
            idx.gain.set_value(60000.0)
        head = mod.connect(idx)
    head.connect(car.frequency)
    if variant == "two-mods":
        m2 = c.create_oscillator(type_="triangle", frequency=3.0)
        m2.connect(c.create_gain(gain=20.0)).connect(car.frequency)
        m2.start()
    if variant == "shared":
        mod.connect(c.create_gain(gain=0.1)).connect(c.destination())
    car.connect(c.destination())
    if variant == "square-mod":
        mod.start_at(700.3 / 48000.0)
        mod.stop_at(frames / 48000.0 * 0.6)
    else:
        mod.start()
    car.start_at(0.0005)
    plan = c.plan_describe() if binding.prefix == "waa_" else ""
    out = None if plan_only else c.start_rendering_sync().data
    c.close()
    return out, plan


FM_VARIANTS = ["plain", "square-mod", "ramped-index", "no-gain", "clamped", "shared", "two-mods"]
FM_NOT_FOLDED = ("ramped-index", "shared", "two-mods")


@pytest.mark.measure
@pytest.mark.parametrize("variant", FM_VARIANTS)
def test_plan_fm_fold(variant):
    _, plan = _two_operator(waa.measure_binding(), 3, 128 * 40, variant, plan_only=True)
    assert ("folded into the carrier's prefix-sum kernel" in plan) == (variant not in FM_NOT_FOLDED), plan


@pytest.mark.measure
@pytest.mark.gpu
@pytest.mark.parametrize("variant", FM_VARIANTS)
def test_parity_fm_fold(hip, orc, variant, monkeypatch):
    """the carrier evaluating modulator, index and mix_to_output itself (OscDesc::fm_*) is the same arithmetic as the three
    launches it stands for: bit-identical to them (WAA_NO_FM_FOLD), and within the oscillator's tolerance of the oracle"""
    n, frames = 3, 128 * 60 + 37
    g, plan = _two_operator(hip, n, frames, variant)
    assert ("folded into the carrier's prefix-sum kernel" in plan) == (variant not in FM_NOT_FOLDED), plan
    o, _ = _two_operator(orc, n, frames, variant)
    assert np.abs(o).max() > 0.5
    assert np.abs(g - o).max() <= 2e-4 and rms_err(g, o).max() <= 2e-5, (float(np.abs(g - o).max()), rms_err(g, o))
    monkeypatch.setenv("WAA_NO_FM_FOLD", "1")
    u, plan = _two_operator(hip, n, frames, variant)
    assert "folded into the carrier" not in plan
    assert np.array_equal(g, u)
