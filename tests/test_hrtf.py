"""PannerNode with the HRTF panning model (src/node/panner.rs:39-68,225-275,697-711,781-829).

The arithmetic is the third-party crate hrtf 0.8.1 (not vendored); the reference asserts only "differs from the input,
non-zero tail" (panner.rs:1226-1269) => sample values are PARITY UNPINNED by the reference.  What pins them here:

* the reference's own HRTF test, re-typed;
* the written definition restated independently in numpy f64: the HRIR pair of a direction is the
  barycentric mix of the three measured HRIRs of the sphere triangle the direction pierces, the output of a render
  quantum is gain * (HRIR of THIS quantum) convolved with the mono input continued into the previously processed quanta;
* the reference's control flow (in the reference): tail counter that is never reset (:697-711), k-rate params (first
  value, :781-799), stereo input mixed down and doubled (:800-810), the y/z swap of the direction (:246-250).
"""
import os
import struct

import numpy as np
import pytest

import web_audio_api_rs_amd as waa

SR = 44100.0
RQ = 128
BIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "IRC_1003_C.bin")


def load_sphere():
    d = open(BIN, "rb").read()
    assert d[:4] == b"HRIR"
    sr, length, nv, ni = struct.unpack("<4I", d[4:20])
    idx = np.frombuffer(d, "<u4", ni, 20).reshape(-1, 3)
    off = 20 + 4 * ni
    pos = np.zeros((nv, 3), np.float32)
    left = np.zeros((nv, length), np.float32)
    right = np.zeros((nv, length), np.float32)
    for v in range(nv):
        pos[v] = np.frombuffer(d, "<f4", 3, off)
        left[v] = np.frombuffer(d, "<f4", length, off + 12)
        right[v] = np.frombuffer(d, "<f4", length, off + 12 + 4 * length)
        off += 12 + 8 * length
    assert off == len(d)
    return sr, idx, pos, left, right


SPHERE = load_sphere()


def sample_bilinear(direction):
    """f64: the face whose plane the ray meets in front and in which the piercing point lies (all weights >= 0)."""
    _, idx, pos, left, right = SPHERE
    d = np.asarray(direction, np.float64)
    best, pick = -1e30, None
    for f in idx:
        a, b, c = (pos[k].astype(np.float64) for k in f)
        n = np.cross(b - a, c - a)
        den = d @ n
        if den == 0:
            continue
        t = (a @ n) / den
        if t <= 0:
            continue
        p = d * t
        m = np.array([b - a, c - a]).T
        vw, *_ = np.linalg.lstsq(m, p - a, rcond=None)
        u = 1 - vw.sum()
        w3 = np.array([u, vw[0], vw[1]])
        if w3.min() > best:
            best, pick = w3.min(), (f, w3)
    f, w3 = pick
    hl = sum(w3[k] * left[f[k]].astype(np.float64) for k in range(3))
    hr = sum(w3[k] * right[f[k]].astype(np.float64) for k in range(3))
    return hl, hr


def direction_of(position, listener=((0, 0, 0), (0, 0, -1), (0, 1, 0))):
    """spatial.rs:205-270 azimuth / elevation -> unit vector -> hrtf's Vec3 {x: p[0], z: p[1], y: p[2]} (f64)."""
    sp = np.asarray(position, np.float64)
    lp, lf, lu = (np.asarray(v, np.float64) for v in listener)
    rel = sp - lp
    if not rel.any():
        return np.array([0.0, 1.0, 0.0])  # azimuth 0, elevation 0: (0, 0, 1) with y/z swapped
    src = rel / np.linalg.norm(rel)
    right = np.cross(lf, lu)
    right /= np.linalg.norm(right)
    fwd = lf / np.linalg.norm(lf)
    up = np.cross(right, fwd)
    proj = src - up * (src @ up)
    proj /= np.linalg.norm(proj)
    az = np.degrees(np.arccos(np.clip(proj @ right, -1, 1)))
    if proj @ fwd < 0:
        az = 360 - az
    az = 90 - az if 0 <= az <= 270 else 450 - az
    el = 90 - np.degrees(np.arccos(np.clip(src @ up, -1, 1)))
    if el > 90:
        el = 180 - el
    elif el < -90:
        el = -180 - el
    a, e = np.radians(az), np.radians(el)
    x, y, z = np.sin(a) * np.cos(e), np.sin(e), np.cos(a) * np.cos(e)
    return np.array([x, z, y])


def definition_render(x_mono, hrirs, gains, processed, corr=1.0):
    """x_mono: the node's mono input per frame; hrirs[q] = (hl, hr); processed[q]: the node runs in quantum q."""
    nq = len(x_mono) // RQ
    taps = len(hrirs[0][0])
    out = np.zeros((2, nq * RQ))
    hist = np.zeros(0)
    for q in range(nq):
        if not processed[q]:
            continue
        blk = np.asarray(x_mono[q * RQ:(q + 1) * RQ], np.float64)
        ext = np.concatenate([np.zeros(max(0, taps - 1 - len(hist))), hist[-(taps - 1):], blk])
        for e in range(2):
            y = np.convolve(ext, hrirs[q][e])[taps - 1:taps - 1 + RQ]
            out[e, q * RQ:(q + 1) * RQ] = y * gains[q] * corr
        hist = np.concatenate([hist, blk])
    return out


def rms(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


def panner_graph(be, noise, length, position=(1.0, 0.0, 0.0), sr=SR, start=0.0, stop=None, **opts):
    ctx = waa.OfflineAudioContext(2, length, sr, n_instances=noise.shape[0], binding=be)
    src = ctx.create_buffer_source()
    src.set_buffer_batch(noise, sr)
    pan = ctx.create_panner(panning_model="HRTF", position=position, **opts)
    src.connect(pan).connect(ctx.destination())
    src.start_at(start)
    if stop is not None:
        src.stop_at(stop)
    return ctx, src, pan


# ---- the reference's test (panner.rs:1226-1269) ----------------------------------------------------------------------
def test_hrtf(be):
    length = RQ * 4
    ones = np.ones((1, 1, RQ), np.float32)
    ctx, _, pan = panner_graph(be, ones, length, position=(0.0, 0.0, 0.0))
    pan.position_x.set_value(1.0)  # sound comes from the right
    out = ctx.start_rendering_sync().data[0]
    assert pan.panning_model == "HRTF"
    assert np.any(np.abs(out[0, :RQ] - 1.0) > 1e-6) and np.any(np.abs(out[1, :RQ] - 1.0) > 1e-6)
    assert np.any(out[0, RQ:2 * RQ] >= 1e-6) and np.any(out[1, RQ:2 * RQ] >= 1e-6)
    # from the right: the right ear is louder
    assert np.abs(out[1]).sum() > 1.2 * np.abs(out[0]).sum()


def test_needs_the_database():
    saved = waa.api._HRTF_DATABASE
    try:
        waa.api._HRTF_DATABASE = None
        waa.api._HRTF_LOADED.clear()
        ctx = waa.OfflineAudioContext(2, RQ, SR, binding=waa.default_binding(), device=waa.PLAN_ONLY)
        ctx.create_panner(panning_model="HRTF")
        with pytest.raises(waa.WaaError):
            ctx.prepare()
    finally:
        waa.api._HRTF_DATABASE = saved


def test_malformed_sphere_headers_are_refused(hip):
    """waa_hrtf_load_sphere validates the caller's header before it sizes anything (round-2 advisor finding): zero vertices /
    faces, an impulse-response length beyond what the FIR kernels hold, sizes that do not add up."""
    import ctypes
    import struct

    def load(blob):
        return hip.hrtf_load_sphere(ctypes.c_char_p(blob), len(blob))

    good = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "IRC_1003_C.bin"), "rb").read()
    rate, length, nv, ni = struct.unpack_from("<4I", good, 4)
    assert (rate, length) == (44100, 512) and nv > 0 and ni % 3 == 0
    for hdr in [(rate, length, 0, 0), (rate, length, 0, ni), (rate, 0, nv, ni), (rate, 100000, nv, ni), (0, length, nv, ni),
                (rate, length, nv, ni + 1), (rate, length, 0xFFFFFFFF, 0xFFFFFFFC)]:
        blob = b"HRIR" + struct.pack("<4I", *hdr) + good[20:]
        assert load(blob) == 1, hdr  # WAA_ERR_INVALID_ARGUMENT
    assert load(good[:-4]) == 1 and load(b"HRIX" + good[4:]) == 1
    assert load(good) == 0  # (and the real database loads again)


# ---- HrirSphere::sample_bilinear ------------------------------------------------------------------------------------
def test_sample_bilinear(be):
    sr, idx, pos, left, right = SPHERE
    assert int(be.hrtf_hrir_length(SR)) == left.shape[1] == 512 or (waa.ensure_hrtf_database(be) is None and
                                                                    int(be.hrtf_hrir_length(SR)) == 512)
    # at a measured direction: that HRIR itself
    for v in (0, 17, 100, 186):
        h = waa.hrtf_sample(be, SR, pos[v])
        assert np.abs(h[0] - left[v]).max() <= 2e-6 and np.abs(h[1] - right[v]).max() <= 2e-6
    # anywhere: the barycentric mix of the pierced triangle (independent f64 search)
    rng = np.random.default_rng(3)
    for _ in range(40):
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        h = waa.hrtf_sample(be, SR, d.astype(np.float32))
        hl, hr = sample_bilinear(d)
        assert np.abs(h[0] - hl).max() <= 5e-6 and np.abs(h[1] - hr).max() <= 5e-6


# ---- the definition ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("position", [(1.0, 0.0, 0.0), (-2.0, 1.5, -0.5), (0.0, 0.0, 3.0), (0.3, -4.0, 0.2)])
def test_static_source_matches_the_definition(be, position):
    nq = 10
    rng = np.random.default_rng(7)
    x = rng.uniform(-1, 1, (2, 1, nq * RQ)).astype(np.float32)
    ctx, _, _ = panner_graph(be, x, nq * RQ - 5, position=position)
    out = ctx.start_rendering_sync().data
    h = sample_bilinear(direction_of(position))
    dist = float(np.linalg.norm(position))
    gain = 1.0 / (1.0 + (max(dist, 1.0) - 1.0))  # inverse model, refDistance 1, rolloff 1 (panner.rs:955-985)
    for i in range(2):
        ref = definition_render(x[i, 0], [h] * nq, [gain] * nq, [True] * nq)[:, :nq * RQ - 5]
        assert rms(out[i, 0], ref[0]) <= 1e-6 and rms(out[i, 1], ref[1]) <= 1e-6


def test_moving_source_switches_the_hrir_per_quantum(be):
    """k-rate: positionX automated, the first value of every quantum decides (panner.rs:781-799)."""
    nq = 12
    rng = np.random.default_rng(8)
    x = rng.uniform(-1, 1, (1, 1, nq * RQ)).astype(np.float32)
    ctx, _, pan = panner_graph(be, x, nq * RQ, position=(0.0, 0.5, -1.0))
    xs = np.linspace(-3.0, 3.0, nq).astype(np.float32)
    pan.position_x.set_block(0, xs)
    out = ctx.start_rendering_sync().data[0]
    hr, gains = [], []
    for q in range(nq):
        p = (float(xs[q]), 0.5, -1.0)
        hr.append(sample_bilinear(direction_of(p)))
        gains.append(1.0 / (1.0 + (max(float(np.linalg.norm(np.float32(p))), 1.0) - 1.0)))
    ref = definition_render(x[0, 0], hr, gains, [True] * nq)
    assert rms(out[0], ref[0]) <= 1e-6 and rms(out[1], ref[1]) <= 1e-6


def test_stereo_input_is_mixed_down_and_doubled(be):
    nq = 8
    rng = np.random.default_rng(12)
    x = rng.uniform(-1, 1, (1, 2, nq * RQ)).astype(np.float32)
    ctx, _, _ = panner_graph(be, x, nq * RQ, position=(0.0, 2.0, 1.0))
    out = ctx.start_rendering_sync().data[0]
    mono = (np.float32(0.5) * (x[0, 0] + x[0, 1])).astype(np.float32)
    h = sample_bilinear(direction_of((0.0, 2.0, 1.0)))
    gain = 1.0 / (1.0 + (float(np.sqrt(np.float32(5.0))) - 1.0))
    ref = definition_render(mono, [h] * nq, [gain] * nq, [True] * nq, corr=2.0)
    assert rms(out[0], ref[0]) <= 1e-6 and rms(out[1], ref[1]) <= 1e-6


# ---- tail and frozen history -------------------------------------------------------------------------------------------
def test_tail_then_silence(be):
    """The source stops after three quanta: the panner keeps processing (zeros) while tail_time_counter < 512, i.e. four
    more quanta, then reports silence."""
    nq = 12
    rng = np.random.default_rng(14)
    x = rng.uniform(-1, 1, (1, 1, 3 * RQ)).astype(np.float32)
    ctx, _, _ = panner_graph(be, x, nq * RQ)
    out = ctx.start_rendering_sync().data[0]
    xin = np.zeros(nq * RQ)
    xin[:3 * RQ] = x[0, 0]
    h = sample_bilinear(direction_of((1.0, 0.0, 0.0)))
    ref = definition_render(xin, [h] * nq, [1.0] * nq, [q < 7 for q in range(nq)])
    assert rms(out[0], ref[0]) <= 1e-6 and rms(out[1], ref[1]) <= 1e-6
    assert np.any(out[:, 6 * RQ:7 * RQ] != 0.0) and np.all(out[:, 7 * RQ:] == 0.0)


@pytest.mark.parametrize("mode", ["explicit", "clamped-max"])
def test_stereo_source_tail_gain_follows_the_quantum_count(be, mode):
    """A stereo source stops after three quanta.  The tail quanta are silent; with channelCountMode explicit (count 2) they
    are still TWO-channel quanta (quantum.rs:532-569), so the mix-down correction (x 2, panner.rs:800-810) applies to the
    tail as well; with the default clamped-max they are mono and the tail comes out at 1 x.  (Round-2 advisor finding: the
    device rendered the explicit case's tail at 1 x.)"""
    nq = 12
    rng = np.random.default_rng(31)
    x = rng.uniform(-1, 1, (1, 2, 3 * RQ)).astype(np.float32)
    opts = dict(channel_count=2, channel_count_mode="explicit") if mode == "explicit" else {}
    ctx, _, _ = panner_graph(be, x, nq * RQ, **opts)
    out = ctx.start_rendering_sync().data[0]
    xin = np.zeros(nq * RQ)
    xin[:3 * RQ] = (np.float32(0.5) * (x[0, 0] + x[0, 1])).astype(np.float32)
    h = sample_bilinear(direction_of((1.0, 0.0, 0.0)))
    corr = [2.0 if (q < 3 or mode == "explicit") else 1.0 for q in range(nq)]
    ref = definition_render(xin, [h] * nq, corr, [q < 7 for q in range(nq)])
    assert rms(out[0], ref[0]) <= 1e-6 and rms(out[1], ref[1]) <= 1e-6
    assert np.any(out[:, 6 * RQ:7 * RQ] != 0.0) and np.all(out[:, 7 * RQ:] == 0.0)


def _two_sources(be, a, b_, length, start_b, stop_b=None, position=(1.0, 0.0, 0.0)):
    ctx = waa.OfflineAudioContext(2, length, SR, n_instances=a.shape[0], binding=be)
    sa, sb = ctx.create_buffer_source(), ctx.create_buffer_source()
    sa.set_buffer_batch(a, SR)
    sb.set_buffer_batch(b_, SR)
    pan = ctx.create_panner(panning_model="HRTF", position=position)
    sa.connect(pan)
    sb.connect(pan)
    pan.connect(ctx.destination())
    sa.start()
    sb.start_at(start_b)
    if stop_b is not None:
        sb.stop_at(stop_b)
    return ctx


def test_tail_counter_is_never_reset(be):
    """Two bursts (two sources: the device takes its exact per-quantum codes path).  The first gap uses the tail counter
    up (four quanta); after the second burst there is NO tail any more, and had there been a third burst its history
    would be the frozen input of the second."""
    nq = 20
    rng = np.random.default_rng(15)
    a = rng.uniform(-1, 1, (2, 1, 2 * RQ)).astype(np.float32)
    b_ = rng.uniform(-1, 1, (2, 1, 3 * RQ)).astype(np.float32)
    out = _two_sources(be, a, b_, nq * RQ, 10 * RQ / SR).start_rendering_sync().data
    h = sample_bilinear(direction_of((1.0, 0.0, 0.0)))
    processed = [q < 6 or 10 <= q < 13 for q in range(nq)]  # 2 + 4 tail quanta, then the second burst without a tail
    for i in range(2):
        xin = np.zeros(nq * RQ)
        xin[:2 * RQ] = a[i, 0]
        xin[10 * RQ:13 * RQ] = b_[i, 0]
        ref = definition_render(xin, [h] * nq, [1.0] * nq, processed)
        assert rms(out[i, 0], ref[0]) <= 1e-6 and rms(out[i, 1], ref[1]) <= 1e-6
        assert np.all(out[i, :, 13 * RQ:] == 0.0)


def test_mono_then_stereo_input(be):
    """Mono source from t = 0, stereo source later: quanta with a stereo input are mixed down and doubled, mono ones are
    not (panner.rs:800-810) — per quantum."""
    nq = 10
    rng = np.random.default_rng(16)
    a = rng.uniform(-1, 1, (1, 1, nq * RQ)).astype(np.float32)
    b_ = rng.uniform(-1, 1, (1, 2, 3 * RQ)).astype(np.float32)
    out = _two_sources(be, a, b_, nq * RQ, 4 * RQ / SR).start_rendering_sync().data[0]
    h = sample_bilinear(direction_of((1.0, 0.0, 0.0)))
    # the FIR history is the mono signal the node saw; the correction factor follows the quantum's channel count
    mono = a[0, 0].astype(np.float32).copy()
    corr = np.ones(nq)
    for q in range(4, 7):
        sl = slice(q * RQ, (q + 1) * RQ)
        l = (a[0, 0, sl] + b_[0, 0, (q - 4) * RQ:(q - 3) * RQ]).astype(np.float32)
        r = (a[0, 0, sl] + b_[0, 1, (q - 4) * RQ:(q - 3) * RQ]).astype(np.float32)
        mono[sl] = np.float32(0.5) * (l + r)
        corr[q] = 2.0
    ref = definition_render(mono, [h] * nq, list(corr), [True] * nq)
    assert rms(out[0], ref[0]) <= 1e-6 and rms(out[1], ref[1]) <= 1e-6


# ---- other sample rates (our own definition of the crate's HRIR resampling step) ------------------------
@pytest.mark.parametrize("sr", [48000.0, 22050.0])
def test_other_sample_rates(be, sr):
    taps = int(be.hrtf_hrir_length(sr)) if waa.ensure_hrtf_database(be) is None else 0
    rate = max(sr, 27000.0)  # panner.rs:46-49
    ratio = rate / 44100.0
    n = 0
    while (n + 1) / ratio - 128.0 < 512 - 257.0 - 1.0 / ratio:
        n += 1
    assert taps == n
    # the resampled HRIR keeps the energy the measured one has below the new band limit (band-limited interpolation)
    _, _, pos, left, _ = SPHERE
    h = waa.hrtf_sample(be, sr, pos[40])
    spec = np.abs(np.fft.rfft(left[40].astype(np.float64), 8192)) ** 2
    freq = np.fft.rfftfreq(8192, 1.0 / 44100.0)
    e0 = float((left[40].astype(np.float64) ** 2).sum()) * spec[freq < 0.95 * min(rate, 44100.0) / 2].sum() / spec.sum()
    e1 = float((h[0].astype(np.float64) ** 2).sum()) / ratio
    assert 0.9 < e1 / e0 < 1.1
    nq = 8
    rng = np.random.default_rng(18)
    x = rng.uniform(-1, 1, (1, 1, nq * RQ)).astype(np.float32)
    # (a direction away from azimuth +-90 degrees, where the reference's f32 acos loses 0.02 degrees)
    position = (1.0, 0.9, 0.1)
    ctx, _, _ = panner_graph(be, x, nq * RQ, position=position, sr=sr)
    out = ctx.start_rendering_sync().data[0]
    hh = waa.hrtf_sample(be, sr, direction_of(position).astype(np.float32)).astype(np.float64)
    gain = 1.0 / (1.0 + (float(np.linalg.norm(np.float32(position))) - 1.0))
    ref = definition_render(x[0, 0], [(hh[0], hh[1])] * nq, [gain] * nq, [True] * nq)
    assert rms(out[0], ref[0]) <= 1e-6 and rms(out[1], ref[1]) <= 1e-6


@pytest.mark.gpu
def test_hrtf_many_instances_sampled(hip, orc):
    """256 contexts x 2 s with per-instance positions and start times on the device, sampled instances vs the oracle."""
    n_inst, nq = 256, 750
    rng = np.random.default_rng(29)
    x = rng.uniform(-1, 1, (n_inst, 1, nq * RQ)).astype(np.float32)
    sample = [0, 3, 130, 255]
    outs = []
    for be, idx in ((hip, None), (orc, sample)):
        noise = x if idx is None else x[idx]
        ctx = waa.OfflineAudioContext(2, nq * RQ, SR, n_instances=noise.shape[0], binding=be)
        src = ctx.create_buffer_source()
        src.set_buffer_batch(noise, SR)
        pan = ctx.create_panner(panning_model="HRTF")
        src.connect(pan).connect(ctx.destination())
        for k in range(noise.shape[0]):
            inst = k if idx is None else idx[k]
            src.start_at((inst % 5) * 0.013, instance=k)
            pan.position_x.set_value(float(np.cos(inst)) * 2.0, instance=k)
            pan.position_z.set_value(float(np.sin(inst)) * 2.0, instance=k)
            pan.position_y.set_value(0.1 * (inst % 11) - 0.5, instance=k)
        outs.append(ctx.start_rendering_sync().data)
    for k, inst in enumerate(sample):
        for c in range(2):
            assert rms(outs[0][inst, c], outs[1][k, c]) <= 1e-6


@pytest.mark.measure
@pytest.mark.gpu
@pytest.mark.parametrize("positions", ["batch", "instance"])
def test_fir_forms_are_bit_identical(hip, positions):
    """hrtf8_kernel (eight frames per lane, both ears packed, HRIR pair from the scalar cache) against hrtf_kernel (four
    frames per lane, HRIR pair interpolated per unit into LDS): same products, same summation order, same bits — with one
    direction for the batch, one per instance, stereo input, gaps and a ragged end"""
    import os
    n_inst, nq = 5, 37  # (37: the last wavefront of hrtf8_kernel holds one unit)
    rng = np.random.default_rng(43)
    x = rng.uniform(-1, 1, (n_inst, 2, 20 * RQ)).astype(np.float32)

    def render():
        ctx = waa.OfflineAudioContext(2, nq * RQ - 3, SR, n_instances=n_inst, binding=hip)
        src = ctx.create_buffer_source()
        src.set_buffer_batch(x, SR)
        pan = ctx.create_panner(panning_model="HRTF", position=(1.0, 0.5, -0.5))
        src.connect(pan).connect(ctx.destination())
        for k in range(n_inst):
            src.start_at(0.004 * k, instance=k)
            if positions == "instance":
                pan.position_x.set_value(1.0 + 0.3 * k, instance=k)
                pan.position_y.set_value(0.2 * k - 0.4, instance=k)
        return ctx.start_rendering_sync().data

    saved = {k: os.environ.pop(k, None) for k in ("WAA_HRTF_V1", "WAA_HRTF_V8", "WAA_HRTF_DYNAMIC", "WAA_HRTF_DIRECT")}
    try:
        os.environ["WAA_HRTF_DIRECT"] = "1"  # (one direction for the batch takes the transform form since round 6: not this test's subject)
        new = render()
        os.environ.pop("WAA_HRTF_DIRECT")
        os.environ["WAA_HRTF_V1"] = "1"
        old = render()
        os.environ.pop("WAA_HRTF_V1")
        os.environ["WAA_HRTF_DYNAMIC"] = "1"
        os.environ["WAA_HRTF_V8"] = "1"
        lds = render()
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    assert np.any(new != 0)
    assert np.array_equal(new, old)
    assert np.array_equal(lds, old)


@pytest.mark.measure
@pytest.mark.gpu
@pytest.mark.parametrize("sr", [44100.0, 48000.0, 32000.0])
@pytest.mark.parametrize("per_context", [False, True], ids=["one-direction", "a-direction-per-context"])
def test_transform_form_against_the_direct_form_and_the_oracle(hip, orc, sr, per_context):
    """round 6, waa_hrtf_fft.hip: PannerNode and AudioListener at rest = one HRIR pair for the whole batch -> partitioned
    overlap-add on 256-point transforms (2 per quantum, both ears in one) instead of taps x 128 x 2 multiply-adds.  Stereo and mono
    inputs, per-instance start times (gaps: skipped quanta, the frozen history), sources that end (the tail), a ragged end, runs
    shorter than the render (the four unstored quanta in front of every run); against hrtf8_kernel on the same device and against
    the oracle's f64-accumulated direct form"""
    import os
    n_inst, nq = 7, 300
    rng = np.random.default_rng(51)
    x = rng.uniform(-1, 1, (n_inst, 2, 90 * RQ + 17)).astype(np.float32)
    y = rng.uniform(-1, 1, (n_inst, 1, 40 * RQ)).astype(np.float32)

    def render(be):
        ctx = waa.OfflineAudioContext(2, nq * RQ - 5, sr, n_instances=n_inst, binding=be)
        a = ctx.create_buffer_source()
        a.set_buffer_batch(x, sr)
        b2 = ctx.create_buffer_source()
        b2.set_buffer_batch(y, sr)
        pan = ctx.create_panner(panning_model="HRTF", position=(-1.5, 0.4, 0.7), ref_distance=0.5)
        if per_context:  # every context its own source position (at rest): one table per context, built on the device
            for k in range(n_inst):
                pan.position_x.set_value(-2.0 + 0.7 * k, instance=k)
                pan.position_z.set_value(1.5 - 0.4 * k, instance=k)
        a.connect(pan)
        b2.connect(pan)
        pan.connect(ctx.destination())
        for k in range(n_inst):
            a.start_at(0.003 * k, instance=k)
            b2.start_at(0.9 + 0.11 * k, instance=k)   # well after `a` has ended and the tail has run out: a second stretch
        if be is hip:
            plan = ctx.plan_describe()
            assert "partitions of 128 taps as 256-point transforms" in plan, plan  # (WAA_HRTF_DIRECT is a launch-time switch)
            assert ("one direction per context" if per_context else "one direction for the whole batch") in plan, plan
        out = ctx.start_rendering_sync().data
        ctx.close()
        return out

    saved = os.environ.pop("WAA_HRTF_DIRECT", None)
    try:
        fft = render(hip)
        os.environ["WAA_HRTF_DIRECT"] = "1"
        direct = render(hip)
    finally:
        os.environ.pop("WAA_HRTF_DIRECT", None)
        if saved is not None:
            os.environ["WAA_HRTF_DIRECT"] = saved
    ref = render(orc)
    assert np.abs(ref).max() > 0.05
    for k in range(n_inst):
        for c in range(2):
            assert rms(fft[k, c], ref[k, c]) <= 1e-6 and rms(direct[k, c], ref[k, c]) <= 1e-6
            assert rms(fft[k, c], direct[k, c]) <= 5e-7
    # Where the direct form's products are exact zeros INSIDE a processed quantum (the HRIR's leading zeros, the input's gaps) the
    # transform form leaves its roundoff (1e-10 ... 2e-8 of full scale; the crate renders with an FFT too).  Quanta the node SKIPS are zeros in
    # both forms: a quantum that is all zero in the direct form, like the four in front of it, is all zero in the transform form.
    assert np.abs(fft[direct == 0]).max() <= 1e-7  # (measured: 2e-8)
    nqq = direct.shape[2] // RQ
    zq = (direct[:, :, :nqq * RQ].reshape(n_inst, 2, nqq, RQ) == 0).all(axis=(1, 3))
    fq = (fft[:, :, :nqq * RQ].reshape(n_inst, 2, nqq, RQ) == 0).all(axis=(1, 3))
    for k in range(n_inst):
        for q in range(4, nqq):
            if zq[k, q - 4:q + 1].all():
                assert fq[k, q], (k, q)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [502, 630])
def test_transform_form_where_a_node_behind_the_panner_decides_on_exact_zeros(hip, orc, seed):
    """round 6, found by the frozen-state fuzz generator on the first build with waa_hrtf_fft.hip: behind the end of the panner's
    tail the direct form puts out exact zeros, the transforms 1e-10 of roundoff — and a DelayNode / BiquadFilterNode behind the
    panner goes silent (= mono) on exactly that (delay.rs:640-668, biquad_filter.rs:775-790): the channel count behind it flipped
    33 quanta into the render (seed 502: 1.5e-3 of full scale; seed 630, a feedback loop: 6e-2).  In such plans the kernel runs its
    exact-zeros form: it follows the last non-zero input frame and puts out 0 behind the response's reach, like the direct sum"""
    from test_fuzz_graphs import build_random_graph
    ch, descr = build_random_graph(hip, seed, frozen=True)
    plan = ch.plan_describe()
    assert "HRTF" in plan and "exact zeros behind the response's reach" in plan, plan
    g = ch.start_rendering_sync().data
    ch.close()
    co, _ = build_random_graph(orc, seed, frozen=True)
    o = co.start_rendering_sync().data
    co.close()
    scale = max(1.0, float(np.abs(o).max()))
    assert max(rms(g[i, c], o[i, c]) for i in range(g.shape[0]) for c in range(g.shape[1])) <= 1e-6 * scale, descr
    assert np.abs(g - o).max() <= 2e-5 * scale, descr


@pytest.mark.gpu
@pytest.mark.parametrize("consumer", ["hrtf", "convolver"])
def test_buffer_that_ends_on_a_quantum_boundary_read_in_place(hip, orc, consumer):
    """frozen-state fuzz seed 38723 (round 6).  An AudioBuffer of exactly 18 quanta in front of an HRTF panner (or a ConvolverNode) is
    read in place by the node's kernel.  The reference renders ONE more quantum behind such a buffer as active — zeros, not the silent
    block: the source's clock reaches the duration by additions of dt and may still compare below it — and in place that quantum was
    the next context's first one: 0.2 of full scale in the panner's tail for every context but the last.  Such a source is copied now"""
    frames, n_inst = 18 * 128, 3
    x = np.random.default_rng(38723).uniform(-1, 1, (n_inst, 1, frames)).astype(np.float32)
    outs = []
    for be in (hip, orc):
        ctx = waa.OfflineAudioContext(2, 40 * 128, 48000.0, n_instances=n_inst, binding=be)
        src = ctx.create_buffer_source()
        src.set_buffer_batch(x, 48000.0)
        if consumer == "hrtf":
            node = ctx.create_panner(panning_model="HRTF", position=(1.0, 0.2, -0.4))
        else:
            node = ctx.create_convolver(buffer=waa.AudioBuffer(np.random.default_rng(1).uniform(-1, 1, (1, 700)).astype(np.float32) * 0.05, 48000.0),
                                        disable_normalization=True)
        src.connect(node).connect(ctx.destination())
        src.start()
        outs.append(ctx.start_rendering_sync().data)
        ctx.close()
    g, o = outs
    assert np.abs(o[:, :, 18 * 128:22 * 128]).max() > 1e-3  # (the tail)
    for k in range(n_inst):
        for c in range(2):
            assert rms(g[k, c], o[k, c]) <= 1e-6, (k, c)


@pytest.mark.measure
@pytest.mark.gpu
def test_exact_zeros_form_has_the_direct_forms_zeros(hip, orc):
    """the exact-zeros form of the transform kernel (a DelayNode behind the panner: a dynamic plan) against the direct form on the same
    device: bursts with gaps longer and shorter than the response, a per-context start; wherever the direct sum is exactly zero
    the transform form is — except on the handful of frames right at an onset (the response's leading zero taps are not modelled)"""
    import os
    n_inst, nq, sr = 5, 120, 48000.0
    rng = np.random.default_rng(77)
    x = np.zeros((n_inst, 2, 80 * RQ), np.float32)  # (the source ENDS inside the render: the DelayNode behind the panner makes the plan dynamic)
    for k in range(n_inst):
        for (a, b2) in ((3, 9), (11, 12), (20, 31), (40, 41), (60, 75)):
            lo, hi = a * RQ + 17 * k, b2 * RQ + 40 + 9 * k
            x[k, :, lo:hi] = rng.uniform(-1, 1, (2, hi - lo))

    def render(be):
        ctx = waa.OfflineAudioContext(2, nq * RQ, sr, n_instances=n_inst, binding=be)
        a = ctx.create_buffer_source()
        a.set_buffer_batch(x, sr)
        a.start()
        pan = ctx.create_panner(panning_model="HRTF", position=(0.8, -0.3, 1.1))
        dl = ctx.create_delay(0.1)
        dl.delay_time.set_value(0.01)
        a.connect(pan).connect(dl).connect(ctx.destination())
        pan.connect(ctx.destination())
        plan = ctx.plan_describe() if be is not orc else ""
        out = ctx.start_rendering_sync().data
        ctx.close()
        return out, plan

    saved = os.environ.pop("WAA_HRTF_DIRECT", None)
    try:
        fft, plan = render(hip)
        assert "exact zeros behind the response's reach" in plan, plan
        os.environ["WAA_HRTF_DIRECT"] = "1"
        direct, _ = render(hip)
    finally:
        os.environ.pop("WAA_HRTF_DIRECT", None)
        if saved is not None:
            os.environ["WAA_HRTF_DIRECT"] = saved
    ref, _ = render(orc)
    for k in range(n_inst):
        for c in range(2):
            assert rms(fft[k, c], ref[k, c]) <= 1e-6 and rms(direct[k, c], ref[k, c]) <= 1e-6
    stray = (direct == 0) & (fft != 0)
    assert np.abs(fft[stray]).max(initial=0.0) <= 1e-7
    assert stray.sum() <= 40 * n_inst * 2 * 5, stray.sum()   # (onsets only: five bursts per context)
    zq_d = (direct.reshape(n_inst, 2, nq, RQ) == 0).all(axis=3)
    zq_f = (fft.reshape(n_inst, 2, nq, RQ) == 0).all(axis=3)
    assert np.array_equal(zq_d, zq_f)   # quanta of exact zeros: the same ones
