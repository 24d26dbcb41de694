"""Graph builders shared by the parity tests (BASELINE.json configs C1..C5, T1 at small sizes)."""
import numpy as np

import web_audio_api_rs_amd as waa


def white_noise(n_inst, n_ch, frames, seed0=0xA0D10, first=0):
    """uniform white noise in [-1, 1), seed = 0xA0D10 + instance."""
    out = np.empty((n_inst, n_ch, frames), np.float32)
    for i in range(n_inst):
        rng = np.random.default_rng(seed0 + first + i)
        out[i] = rng.uniform(-1.0, 1.0, (n_ch, frames)).astype(np.float32)
    return out


_GARAGE_CACHE = {}


def garage_ir(binding=None, sr=48000.0):
    """The impulse response BASELINE.json config 3 names: samples/parking-garage-response.wav of the reference
    (2 ch, 44.1 kHz, 16-bit PCM, 164 363 frames; committed as tests/golden/parking-garage-response.wav), decoded the
    way the reference's decoder does (symphonia's i16 -> f32 conversion, third party, restated: sample / 32768) and
    resampled to the context rate by AudioBuffer::resample (decoding.rs:51 -> buffer.rs:311-363): 44 100 -> 48 000 Hz
    gives 178 899 frames = 175 reference partitions of 1024.  `binding` selects whose
    resampler runs (the product's waa_buffer_resample by default; tests check it against the oracle's bit for bit)."""
    import os
    import wave
    if binding is None:
        binding = waa.default_binding()
    key = (binding.prefix if hasattr(binding, "prefix") else id(binding), float(sr))
    if key not in _GARAGE_CACHE:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parking-garage-response.wav")
        with wave.open(path) as w:
            assert (w.getnchannels(), w.getframerate(), w.getsampwidth(), w.getnframes()) == (2, 44100, 2, 164363)
            raw = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").reshape(-1, 2).T
        pcm = np.ascontiguousarray(raw.astype(np.float32) / np.float32(32768.0))
        ir = pcm if float(sr) == 44100.0 else waa.resample(binding, pcm, 44100.0, sr)
        if float(sr) == 48000.0:
            assert ir.shape == (2, 178899), ir.shape
        _GARAGE_CACHE[key] = ir
    return _GARAGE_CACHE[key]


def garage_like_ir(frames=178899, n_ch=2, sr=48000.0, seed=7):
    """Synthetic stand-in for samples/parking-garage-response.wav resampled to 48 kHz (2 ch x 178 899
    frames => 175 partitions of 1024): exponentially decaying noise, ~1.2 s RT60-ish."""
    rng = np.random.default_rng(seed)
    t = np.arange(frames, dtype=np.float64) / sr
    # decaying reverb tail over a noise floor (a recorded IR never decays to digital silence): after the
    # reference's normalisation the last samples stay above FFTConvolver's 1e-6 trim threshold, so all 175
    # reference partitions (22 device blocks of 8192) are exercised
    env = np.exp(-t / 0.35) + 2e-3
    ir = (rng.uniform(-1.0, 1.0, (n_ch, frames)) * env).astype(np.float32)
    ir[:, -1] = np.float32(0.5 * 2e-3)
    return ir


def c2(binding, noise, sr=48000.0, length=None, ftype="lowpass", freq=200.0, q=1.0, gain=0.5, device=-1):
    """C2: src -> Biquad(lowpass 200 Hz, Q 1) -> Gain(0.5) -> destination."""
    n_inst, n_ch, frames = noise.shape
    ctx = waa.OfflineAudioContext(2, length or frames, sr, n_instances=n_inst, binding=binding, device=device)
    src = ctx.create_buffer_source()
    src.set_buffer_batch(noise, sr)
    flt = ctx.create_biquad_filter(type_=ftype, frequency=freq, q=q)
    g = ctx.create_gain(gain=gain)
    src.connect(flt).connect(g).connect(ctx.destination())
    src.start()
    return ctx, dict(src=src, biquad=flt, gain=g)


def c5(binding, noise, sr=48000.0, length=None, buf_sr=None, rate=1.5, loop=True):
    """C5: src(playbackRate != 1 or foreign buffer rate, loop) -> WaveShaper(2048-pt cos curve) -> destination."""
    n_inst, n_ch, frames = noise.shape
    ctx = waa.OfflineAudioContext(2, length or frames, sr, n_instances=n_inst, binding=binding)
    src = ctx.create_buffer_source()
    src.set_buffer_batch(noise, buf_sr or sr)
    src.playback_rate.set_value(rate)
    src.set_loop(loop)
    i = np.arange(2048, dtype=np.float32)
    curve = np.cos(np.float32(np.pi) + i * np.float32(np.pi) / np.float32(2047)).astype(np.float32)  # waveshaper.rs:78-87
    sh = ctx.create_wave_shaper(curve=curve)
    src.connect(sh).connect(ctx.destination())
    src.start()
    return ctx, dict(src=src, shaper=sh)


def t1(binding, noise, ir, sr=48000.0, length=None, with_biquad=True, device=-1, biquad_handle=False):
    """T1 / C3: src -> [Biquad] -> Convolver(IR, normalize) -> destination."""
    n_inst, n_ch, frames = noise.shape
    ctx = waa.OfflineAudioContext(2, length or frames, sr, n_instances=n_inst, binding=binding, device=device)
    src = ctx.create_buffer_source()
    src.set_buffer_batch(noise, sr)
    conv = ctx.create_convolver(buffer=waa.AudioBuffer(ir, sr))
    node = src
    nodes = dict(src=src, conv=conv)
    if with_biquad:
        nodes["biquad"] = ctx.create_biquad_filter(type_="lowpass", frequency=200.0, q=1.0)
        node = src.connect(nodes["biquad"])
    node.connect(conv).connect(ctx.destination())
    src.start()
    return ctx, nodes


def c4(binding, noise, ir, sr=48000.0, length=None, device=-1):
    """C4: src -> Biquad -> Convolver -> StereoPanner(0.1) -> Analyser(2048, 0.8) -> destination."""
    n_inst, n_ch, frames = noise.shape
    ctx = waa.OfflineAudioContext(2, length or frames, sr, n_instances=n_inst, binding=binding, device=device)
    src = ctx.create_buffer_source()
    src.set_buffer_batch(noise, sr)
    flt = ctx.create_biquad_filter(type_="lowpass", frequency=200.0, q=1.0)
    conv = ctx.create_convolver(buffer=waa.AudioBuffer(ir, sr))
    pan = ctx.create_stereo_panner(pan=0.1)
    an = ctx.create_analyser(fft_size=2048, smoothing_time_constant=0.8)
    src.connect(flt).connect(conv).connect(pan).connect(an).connect(ctx.destination())
    src.start()
    return ctx, dict(src=src, analyser=an)


def rms_err(a, b):
    """per (instance, channel) RMS error, f64"""
    return np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2, axis=-1))


def assert_all_finite(a, what):
    """A render (or a pulled array) with a NaN / inf anywhere is a failure by itself: comparisons written as `x > worst` or
    Python's max(worst, x) DROP a NaN (nan > x is False, max(0.0, nan) is 0.0), so every every-instance helper calls this on
    the device's array first."""
    a = np.asarray(a)
    bad = ~np.isfinite(a)
    if bad.any():
        k = np.unravel_index(int(np.argmax(bad)), a.shape)
        raise AssertionError(f"{what}: {int(bad.sum())} non-finite values, the first at index {tuple(int(i) for i in k)} = {a[k]}")


def strict_max(*values):
    """max() that FAILS on a NaN instead of dropping it (the accumulator form of the every-instance tests)"""
    vals = [float(v) for v in values]
    for v in vals:
        if v != v:
            raise AssertionError(f"NaN in an error accumulation: {vals}")
    return max(vals)


def assert_le(x, tol, what=""):
    """x <= tol, a NaN on either side fails"""
    x = float(x)
    assert x == x and x <= tol, (what, x, tol)
