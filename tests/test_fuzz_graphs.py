"""Randomised graph parity: seeded random Web Audio graphs (sources, filters, panners, delays, fan-in / fan-out,
AudioParam modulation, feedback loops through DelayNodes) rendered by the HIP path and by the oracle.
This is the planner's safety net: processing order, summing order, channel-count propagation, materialisation,
chain fusion, loop scheduling and param chains all have to agree with the reference's per-quantum graph walk.

A graph the device path refuses with status 4 (out of scope, e.g. a biquad on a 4-channel signal) is skipped —
refusing loudly is the contract; mis-rendering is the failure."""
import os

import numpy as np
import pytest
from scipy import signal

import web_audio_api_rs_amd as waa
from graphs import rms_err, white_noise

RQ = 128
SR = 48000.0
N_INST = int(os.environ.get("FUZZ_INST", 3))
# Sources that start late change the reference's DYNAMIC channel counts mid-render (a silent input is mono); the
# device plan uses static counts and says so in the plan; such graphs are skipped below.
LATE_STARTS = True
FRAMES = int(os.environ.get("FUZZ_FRAMES", 2048 * 5 + 200))   # (FUZZ_FRAMES / FUZZ_INST: campaign variants, tools/fuzz_campaign.py)


def build_random_graph(be, seed, frozen=False, tap=None):
    """frozen: WaveShapers may oversample (2x / 4x) and PannerNodes may use the HRTF model (extra draws: other graphs
    than the same seed without it).  tap = k (debugging aid, tools/fuzz_tap_probe.py): the SAME graph, but only node k of
    its node list feeds the destination — where along the graph does a difference start?"""
    rng = np.random.default_rng(seed if not frozen else seed + 100000)
    # FUZZ_MIXED_COUNTS=1 (campaign variant): one instance of some buffer sources plays an AudioBuffer of another channel
    # count (own generator: the graphs of a seed stay what they are without the switch)
    mix_rng = np.random.default_rng(seed + 7000003) if os.environ.get("FUZZ_MIXED_COUNTS") else None
    # FUZZ_WIDE=1 (campaign variant, round 6): AudioBuffers of 1 / 2 / 4 / 6 / 8 channels, destinations of 2 ... 8 channels with either
    # interpretation, GainNodes with wide explicit counts — the speakers table, discrete mixing above six channels and the ORDER in
    # which a node's inputs are summed (own generator: the graphs of a seed stay what they are without the switch)
    wide_rng = np.random.default_rng(seed + 9000011) if os.environ.get("FUZZ_WIDE") else None
    n_out = int(wide_rng.choice([2, 2, 4, 6, 8])) if wide_rng is not None else 2
    c = waa.OfflineAudioContext(n_out, FRAMES, SR, n_instances=N_INST, binding=be)
    if wide_rng is not None and wide_rng.random() < 0.3:
        c.destination().set_channel_interpretation("discrete")
    outputs = []      # nodes that can feed others
    descr = []

    def add_source():
        kind = rng.choice(["buffer1", "buffer2", "buffer2", "constant", "osc"])
        if os.environ.get("FUZZ_STEREO_ONLY"):
            kind = "buffer2"
        if kind.startswith("buffer"):
            nch = int(kind[-1])
            if wide_rng is not None and wide_rng.random() < 0.6:
                nch = int(wide_rng.choice([1, 2, 4, 6, 8]))
            n = c.create_buffer_source()
            length = FRAMES if rng.random() < 0.7 else int(rng.integers(300, FRAMES // 2))  # some end early
            n.set_buffer_batch(white_noise(N_INST, nch, length, seed0=int(rng.integers(1, 1 << 20))) * 0.5, SR)
            if mix_rng is not None and mix_rng.random() < 0.4:
                other = int(mix_rng.choice([k for k in (1, 2, 4) if k != nch]))
                n.set_buffer(waa.AudioBuffer(white_noise(1, other, length, seed0=int(mix_rng.integers(1, 1 << 20)))[0] * 0.5, SR),
                             instance=int(mix_rng.integers(0, N_INST)))
            r = rng.random()
            if LATE_STARTS and r < 0.2:
                n.start_at(float(rng.integers(0, 600)) / SR)
            elif LATE_STARTS and r < 0.3:  # per-instance start times
                for i in range(N_INST):
                    n.start_at(float(rng.integers(0, 600)) / SR, instance=i)
            else:
                n.start()
            if rng.random() < 0.15:  # k-rate playbackRate / detune automation: the slow track with a changing rate
                n.playback_rate.set_value_at_time(1.0, 0.0).linear_ramp_to_value_at_time(float(rng.uniform(0.5, 1.8)),
                                                                                          FRAMES / SR * 0.8)
            elif rng.random() < 0.1:
                n.detune.set_value_at_time(float(rng.uniform(-700.0, 700.0)), FRAMES / SR * float(rng.uniform(0.1, 0.6)))
        elif kind == "constant":
            n = c.create_constant_source(offset=float(rng.uniform(-0.5, 0.5)))
            n.start_at(float(rng.integers(0, 300)) / SR if LATE_STARTS else 0.0)
            if rng.random() < 0.3:
                n.stop_at(float(rng.integers(2000, FRAMES)) / SR)
            if rng.random() < 0.3:  # automated offset (a-rate)
                n.offset.set_target_at_time(float(rng.uniform(-0.5, 0.5)), FRAMES / SR * 0.25, float(rng.uniform(0.005, 0.05)))
        else:
            n = c.create_oscillator(type_=str(rng.choice(["sine", "triangle", "sawtooth", "square"])),
                                    frequency=float(rng.uniform(50.0, 2000.0)))
            if rng.random() < 0.2:  # custom PeriodicWave (periodic_wave.rs:35-190)
                k = int(rng.integers(2, 9))
                n.set_periodic_wave(c.create_periodic_wave(real=rng.uniform(-1, 1, k).astype(np.float32),
                                                           imag=rng.uniform(-1, 1, k).astype(np.float32),
                                                           disable_normalization=bool(rng.integers(0, 2))))
            r = rng.random()
            if LATE_STARTS and r < 0.2:  # sub-quantum start, stop before the end
                n.start_at(float(rng.integers(0, 700)) / SR)
                n.stop_at(float(rng.integers(min(3000, FRAMES - 1), FRAMES)) / SR)   # (short FUZZ_FRAMES variants)
            else:
                n.start()
            if rng.random() < 0.3:  # a glide: a-rate frequency (prefix-sum phase kernel)
                n.frequency.exponential_ramp_to_value_at_time(float(rng.uniform(100.0, 3000.0)), FRAMES / SR * 0.9)
            elif rng.random() < 0.15:
                n.detune.set_value_at_time(0.0, 0.0).linear_ramp_to_value_at_time(float(rng.uniform(-1200.0, 1200.0)), FRAMES / SR)
        outputs.append(n)
        descr.append(kind)
        return n

    def add_processor():
        kind = str(rng.choice(["gain", "gain", "biquad", "biquad", "iir", "shaper", "pan", "delay", "delay", "conv",
                               "panner", "analyser", "cfg-gain", "krate-gain", "krate-biquad", "auto-gain", "auto-biquad",
                               "auto-pan", "auto-delay"]))
        nq = (FRAMES + RQ - 1) // RQ
        if kind == "gain":
            n = c.create_gain(gain=float(rng.uniform(-1.0, 1.0)))
            if rng.random() < 0.3:  # per-instance values
                for i in range(N_INST):
                    n.gain.set_value(float(rng.uniform(-1.0, 1.0)), instance=i)
        elif kind == "cfg-gain":  # explicit / clamped-max channel configs, discrete interpretation
            cc, mode, interp = [(1, "explicit", "speakers"), (2, "explicit", "speakers"), (1, "clamped-max", "speakers"),
                                (2, "explicit", "discrete"), (4, "explicit", "discrete")][int(rng.integers(0, 5))]
            if wide_rng is not None and wide_rng.random() < 0.6:
                cc, mode, interp = [(6, "explicit", "speakers"), (8, "explicit", "speakers"), (4, "clamped-max", "speakers"),
                                    (6, "clamped-max", "discrete"), (4, "explicit", "speakers"), (7, "explicit", "discrete")][int(wide_rng.integers(0, 6))]
            n = c.create_gain(gain=float(rng.uniform(0.2, 1.0)), channel_count=cc, channel_count_mode=mode,
                              channel_interpretation=interp)
        elif kind == "krate-gain":  # one value per render quantum (includes exact 0 and 1: gain.rs fast paths)
            n = c.create_gain(gain=0.5)
            vals = rng.uniform(-1.0, 1.0, nq).astype(np.float32)
            vals[3::11] = 1.0
            if rng.random() < 0.25:
                vals[::7] = 0.0  # a zero gain emits a SILENT (mono) quantum: the planner must flag it
            n.gain.set_block(0, vals)
        elif kind == "auto-gain":  # AudioParam automation methods (param.rs:796-1584), evaluated by the timeline
            n = c.create_gain(gain=float(rng.uniform(0.1, 0.9)))
            t_end = FRAMES / SR
            style = int(rng.integers(0, 4))
            if style == 0:
                n.gain.set_value_at_time(0.1, 0.0).linear_ramp_to_value_at_time(float(rng.uniform(0.3, 1.0)), t_end * 0.7)
            elif style == 1:
                n.gain.set_value_at_time(0.8, t_end * 0.1).exponential_ramp_to_value_at_time(0.05, t_end * 0.9)
            elif style == 2:
                n.gain.set_target_at_time(float(rng.uniform(0.0, 1.0)), t_end * 0.2, float(rng.uniform(0.005, 0.05)))
                n.gain.cancel_and_hold_at_time(t_end * 0.6)
            else:
                n.gain.set_value_curve_at_time(rng.uniform(0.0, 1.0, int(rng.integers(2, 9))).astype(np.float32), t_end * 0.15,
                                               t_end * 0.5)
        elif kind == "auto-biquad":  # a-rate frequency: per-frame coefficients (biquad_filter.rs:837-855)
            n = c.create_biquad_filter(type_=str(rng.choice(["lowpass", "bandpass", "highshelf"])), frequency=300.0,
                                       q=float(rng.uniform(0.5, 3.0)), gain=float(rng.uniform(-6.0, 6.0)))
            n.frequency.set_value_at_time(200.0, 0.0).exponential_ramp_to_value_at_time(float(rng.uniform(1000.0, 9000.0)),
                                                                                        FRAMES / SR)
        elif kind == "auto-delay":  # a chorus-like sweep of delayTime (a-rate, delay.rs:591-606)
            n = c.create_delay(0.05, delay_time=0.01)
            n.delay_time.set_value_at_time(0.004, 0.0).linear_ramp_to_value_at_time(float(rng.uniform(0.01, 0.04)), FRAMES / SR)
        elif kind == "auto-pan":
            n = c.create_stereo_panner(pan=0.0)
            n.pan.set_value_at_time(-1.0, 0.0).linear_ramp_to_value_at_time(1.0, FRAMES / SR * float(rng.uniform(0.5, 1.0)))
        elif kind == "krate-biquad":
            n = c.create_biquad_filter(type_="lowpass", frequency=1000.0, q=1.0)
            n.frequency.set_block(0, np.geomspace(200.0, 6000.0, nq).astype(np.float32))
        elif kind == "panner":
            n = c.create_panner(position=tuple(float(v) for v in rng.uniform(-3.0, 3.0, 3)),
                                distance_model=str(rng.choice(["inverse", "linear", "exponential"])),
                                ref_distance=float(rng.uniform(0.5, 2.0)),
                                panning_model="HRTF" if frozen and rng.random() < 0.6 else "equalpower")
            if frozen and rng.random() < 0.4:  # a moving source: the HRIR pair changes from quantum to quantum
                n.position_x.set_block(0, np.linspace(-3.0, 3.0, nq).astype(np.float32))
        elif kind == "analyser":
            n = c.create_analyser(fft_size=256)
        elif kind == "biquad":
            n = c.create_biquad_filter(type_=str(rng.choice(["lowpass", "highpass", "bandpass", "peaking", "notch"])),
                                       frequency=float(rng.uniform(100.0, 8000.0)), q=float(rng.uniform(0.3, 4.0)),
                                       gain=float(rng.uniform(-6.0, 6.0)))
        elif kind == "iir":
            b, a = signal.butter(int(rng.integers(1, 5)), float(rng.uniform(0.1, 0.6)))
            n = c.create_iir_filter(b, a)
        elif kind == "shaper":
            n = c.create_wave_shaper(curve=np.tanh(np.linspace(-2.0, 2.0, int(rng.choice([3, 64, 257])))).astype(np.float32))
            if frozen:
                n.set_oversample(str(rng.choice(["none", "2x", "4x", "2x"])))
                if rng.random() < 0.25:  # a curve that does not map 0 to 0: silence is processed, mono
                    n.curve = (n.curve + np.float32(0.2)).astype(np.float32)
        elif kind == "pan":
            n = c.create_stereo_panner(pan=float(rng.uniform(-1.0, 1.0)))
        elif kind == "delay":
            n = c.create_delay(0.1, delay_time=float(rng.choice([0.0, 0.0007, 0.003, 0.01, 0.05, 0.09])))
            if rng.random() < 0.3:  # per-instance delay times (different loop strategies per batch are not possible:
                for i in range(N_INST):  # the planner must take the most restrictive one)
                    n.delay_time.set_value(float(rng.choice([0.0, 0.002, 0.03, 0.06])), instance=i)
        else:
            ir = (rng.uniform(-1, 1, (int(rng.choice([1, 2])), int(rng.choice([16, 100, 700])))) *
                  np.exp(-np.arange(1)[None, :])).astype(np.float32)
            n = c.create_convolver(buffer=waa.AudioBuffer(ir, SR))
        descr.append(kind)
        return n

    for _ in range(int(rng.integers(1, 4))):
        add_source()
    procs = []
    for _ in range(int(rng.integers(2, 8))):
        n = add_processor()
        # 1..3 inputs from anything created so far (forward edges only: a DAG)
        for src in rng.choice(len(outputs), size=min(len(outputs), int(rng.integers(1, 4))), replace=False):
            outputs[int(src)].connect(n)
        outputs.append(n)
        procs.append(n)
    # feedback: a later node back into an earlier DelayNode (the delay breaks the cycle), attenuated
    delays = [p for p in procs if isinstance(p, waa.DelayNode)]
    if delays and rng.random() < 0.6:
        d = delays[int(rng.integers(0, len(delays)))]
        later = [p for p in procs[procs.index(d):] if not isinstance(p, (waa.ConvolverNode, waa.IIRFilterNode))]
        tail = later[int(rng.integers(0, len(later)))]
        fb = c.create_gain(gain=float(rng.uniform(-0.5, 0.5)))
        tail.connect(fb).connect(d)
        descr.append("feedback")
        # FUZZ_LOOP_PARAM=1 (campaign variant, round 6): the feedback gain is driven by the loop's own signal — an AudioParam modulated
        # from inside its feedback loop (own generator: the graphs of a seed stay what they are without the switch)
        if os.environ.get("FUZZ_LOOP_PARAM"):
            lp_rng = np.random.default_rng(seed + 5000011)
            if lp_rng.random() < 0.7:
                # (through a limiter: a gain that follows the loop's signal without bound makes the loop quadratic — random graphs of that
                # kind blow up to 1e37 within the render, on both back-ends at the same frame, and compare nothing)
                lim = c.create_wave_shaper(curve=np.tanh(np.linspace(-3.0, 3.0, 65)).astype(np.float32))
                depth = c.create_gain(gain=float(lp_rng.uniform(-0.1, 0.1)))
                (d if lp_rng.random() < 0.5 else tail).connect(lim).connect(depth)
                depth.connect(fb.gain)
                descr.append("loop-param")
    # audio-rate modulation of a param from a source
    gains = [p for p in procs if isinstance(p, waa.GainNode)]
    if gains and rng.random() < 0.5:
        lfo = c.create_oscillator(type_="sine", frequency=float(rng.uniform(1.0, 20.0)))
        depth = c.create_gain(gain=float(rng.uniform(0.05, 0.4)))
        lfo.connect(depth).connect(gains[0].gain)
        lfo.start()
        descr.append("param-mod")
    # everything without a consumer goes to the destination, plus one random extra tap
    fed = {e[0] for e in c._edges}
    extra = int(rng.integers(0, len(outputs)))
    if tap is not None:
        if tap >= len(outputs):
            return None, "+".join(descr)
        outputs[tap].connect(c.destination())
        return c, "+".join(descr) + " | tap %d = %s" % (tap, type(outputs[tap]).__name__)
    for n in outputs:
        if n.id not in fed:
            n.connect(c.destination())
    outputs[extra].connect(c.destination())
    return c, "+".join(descr)


@pytest.mark.measure
@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("FUZZ_FIRST", "0")),
                                        int(os.environ.get("FUZZ_FIRST", "0")) + int(os.environ.get("FUZZ_SEEDS", "60"))))
def test_random_graph_parity(hip, orc, seed):
    ch, descr = build_random_graph(hip, seed)
    try:
        plan = ch.plan_describe()
        g = ch.start_rendering_sync().data
    except waa.WaaError as e:
        if e.status == 4:
            pytest.skip(f"out of scope on the device path: {e} [{descr}]")
        raise
    ch.close()
    assert "dynamic channel count" not in plan  # (the round-1 note of the static plan; WAA_STATIC_CHANNEL_COUNTS only)
    co, _ = build_random_graph(orc, seed)
    o = co.start_rendering_sync().data
    co.close()
    assert np.isfinite(o).all(), descr
    scale = max(1.0, float(np.abs(o).max()))
    err = np.abs(g - o).max()
    assert rms_err(g, o).max() <= 1e-6 * scale, f"{descr}: rms {rms_err(g, o).max():.3g}"
    assert err <= 2e-5 * scale, f"{descr}: max |d| {err:.3g}"


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("FUZZ_FIRST", "0")),
                                        int(os.environ.get("FUZZ_FIRST", "0")) + int(os.environ.get("FUZZ_SEEDS_FROZEN", "40"))))
def test_random_graph_parity_frozen_state_nodes(hip, orc, seed):
    """The same generator with oversampled WaveShapers and HRTF panners mixed in: their frozen
    state over silent quanta, the resamplers' re-creation on a channel-count change and the HRTF tail counter all ride on
    the per-quantum codes of whatever graph surrounds them."""
    ch, descr = build_random_graph(hip, seed, frozen=True)
    try:
        g = ch.start_rendering_sync().data
    except waa.WaaError as e:
        if e.status == 4:
            pytest.skip(f"out of scope on the device path: {e} [{descr}]")
        raise
    ch.close()
    co, _ = build_random_graph(orc, seed, frozen=True)
    o = co.start_rendering_sync().data
    co.close()
    assert np.isfinite(o).all(), descr
    scale = max(1.0, float(np.abs(o).max()))
    assert rms_err(g, o).max() <= 1e-6 * scale, f"{descr}: rms {rms_err(g, o).max():.3g}"
    assert np.abs(g - o).max() <= 2e-5 * scale, f"{descr}: max |d| {np.abs(g - o).max():.3g}"


def test_random_graphs_plan_on_cpu(hip):
    """the same graphs go through the planner without a device (plan-only batches): no crash, a plan or status 4"""
    planned = 0
    for seed in range(60):
        rng_state = seed

        def mk(be):
            return build_random_graph(be, rng_state)

        c, descr = mk(hip)
        c.device = waa.PLAN_ONLY
        try:
            text = c.plan_describe()
            planned += 1
            assert "batch:" in text
        except waa.WaaError as e:
            assert e.status == 4, f"{descr}: {e}"
        c.close()
    assert planned >= 40


@pytest.mark.measure
@pytest.mark.gpu
def test_dynamic_channel_count_is_rendered(hip, orc, monkeypatch):
    """A mono oscillator from t = 0 plus a stereo buffer that starts later, into a BiquadFilter: the reference
    filters ONE channel until the stereo source starts and then starts channel 1 from a zero state
    (biquad_filter.rs:800-815).  The planner's replay finds the count change and renders the graph with exact
    per-quantum channel counts (waa_dyn.hip); WAA_STATIC_CHANNEL_COUNTS=1 brings back the static plan of round 1, which
    filters two channels throughout, says so in the plan, and is refused under WAA_STRICT_CHANNEL_COUNTS."""
    def build(be):
        c = waa.OfflineAudioContext(2, FRAMES, SR, n_instances=1, binding=be)
        osc = c.create_oscillator(frequency=220.0)
        buf = c.create_buffer_source()
        buf.set_buffer_batch(white_noise(1, 2, FRAMES, seed0=3), SR)
        bq = c.create_biquad_filter(type_="lowpass", frequency=300.0)
        osc.connect(bq)
        buf.connect(bq)
        bq.connect(c.destination())
        osc.start()
        buf.start_at(1000.0 / SR)
        return c
    o = build(orc).start_rendering_sync().data
    c = build(hip)
    plan = c.plan_describe()
    assert "dynamic-count group" in plan and "dynamic channel count" not in plan
    g = c.start_rendering_sync().data
    c.close()
    assert rms_err(g, o).max() <= 1e-6 and np.abs(g - o).max() <= 2e-6
    monkeypatch.setenv("WAA_STATIC_CHANNEL_COUNTS", "1")
    c = build(hip)
    assert "dynamic channel count" in c.plan_describe()
    g = c.start_rendering_sync().data
    c.close()
    # the static plan: identical until the stereo source starts, different afterwards (channel 1's filter state)
    assert np.abs(g[:, :, :896] - o[:, :, :896]).max() <= 1e-6
    assert np.abs(g[:, 1, 1024:] - o[:, 1, 1024:]).max() > 1e-4
    monkeypatch.setenv("WAA_STRICT_CHANNEL_COUNTS", "1")
    c = build(hip)
    with pytest.raises(waa.WaaError) as ei:
        c.start_rendering_sync()
    assert ei.value.status == 4


@pytest.mark.gpu
def test_random_graphs_on_poisoned_device_memory(tmp_path):
    """WAA_POISON_ALLOC=1 fills every fresh device allocation with 0xFF bytes (NaN as f32 / f64): a kernel whose output depends
    on memory nobody wrote fails every time instead of once in 20 000 graphs.  300 seeds of each generator in a subprocess
    (tools/fuzz_campaign.py): no NaN output, no mismatch that a second render does not repeat.  (Found in round 3: the
    oscillator fold behind an aliasing AnalyserNode.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "poison.json")
    env = dict(os.environ, WAA_POISON_ALLOC="1")
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "fuzz_campaign.py"), "--first", "700000", "--count", "300",
                           "--jobs", "2", "--out", out], env=env, stdout=subprocess.DEVNULL)
    rec = json.load(open(out))
    for gen, t in rec["generators"].items():
        assert not t["errors"], (gen, t["errors"][:2])
        for m in t["mismatch"]:
            assert np.isfinite(m["max"]) and m["second_render_equals_first"], (gen, m)


@pytest.mark.gpu
def test_dynamic_group_that_needs_more_than_64k_of_lds(hip, orc):
    """fuzz seed 903488 of the round-4 campaign (frozen-state generator): 13 items per quantum on 4-channel signals take
    66.6 KB of dynamic LDS in dyn_kernel<6>, which also has 336 bytes of static LDS — raising the kernel's limit to the full
    160 KB failed silently and the launch was refused ("invalid argument"); the limit is now raised to 160 KB minus the static
    part"""
    ch, descr = build_random_graph(hip, 903488, frozen=True)
    assert "dynamic-count group: 13 item(s)" in ch.plan_describe()
    g = ch.start_rendering_sync().data
    ch.close()
    co, _ = build_random_graph(orc, 903488, frozen=True)
    o = co.start_rendering_sync().data
    co.close()
    scale = max(1.0, float(np.abs(o).max()))
    assert rms_err(g, o).max() / scale <= 1e-6 and np.abs(g - o).max() / scale <= 2e-5
