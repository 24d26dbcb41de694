#!/usr/bin/env python
"""bench.py — throughput of the batched offline render hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c2|t1|c3|c5] [--instances I] [--seconds S]

A "step" = one start_rendering_sync-equivalent pass over one batch of synthetic input:
`instances` independent OfflineAudioContexts x `seconds` s @ 48 kHz stereo, inputs already
resident in HBM (white noise generated on the device).  Default workload = BASELINE.json
configs[1] (C2): 1024 contexts, BufferSource -> Biquad(lowpass 200 Hz, Q 1) -> Gain(0.5) -> destination.

N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ...`, one rank per GPU;
independent batches shard over the GPUs with no data-path collective (weak scaling: every GPU
renders `instances` contexts).  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

SR = 48000.0
RQ = 128


def build_workload(waa, binding, name, n_inst, frames, device, noise_ptr):
    ctx = waa.OfflineAudioContext(2, frames, SR, n_instances=n_inst, binding=binding, device=device)
    if name == "fm":  # two-operator FM: Oscillator -> Gain(index) -> carrier.frequency; carrier -> Gain -> destination
        mod = ctx.create_oscillator(type_="sine", frequency=110.0)
        idx = ctx.create_gain(gain=300.0)
        car = ctx.create_oscillator(type_="sine", frequency=440.0)
        mod.connect(idx).connect(car.frequency)
        car.connect(ctx.create_gain(gain=0.5)).connect(ctx.destination())
        mod.start()
        car.start()
        return ctx, car
    if name == "osc":  # subtractive voice, Oscillator(sawtooth) -> Biquad(lowpass) -> Gain
        osc = ctx.create_oscillator(type_="sawtooth", frequency=110.0)
        for i in range(0, n_inst, max(1, n_inst // 64)):
            osc.detune.set_value(float(i % 1200), instance=i)
        osc.connect(ctx.create_biquad_filter(type_="lowpass", frequency=1200.0, q=2.0)).connect(
            ctx.create_gain(gain=0.5)).connect(ctx.destination())
        osc.start()
        return ctx, osc
    src = ctx.create_buffer_source()
    if noise_ptr is not None:
        src.adopt_device_buffer(noise_ptr, 2, frames, SR)
    node = src
    if name in ("c2", "c2k", "c1a", "t1", "c4"):
        bq = ctx.create_biquad_filter(type_="lowpass", frequency=200.0, q=1.0)
        if name == "c1a":  # BASELINE config 1, a-rate variant (examples/biquad.rs:39-42): 10 Hz -> 10 kHz over 10 s
            bq.frequency.set_value_at_time(10.0, 0.0)
            bq.frequency.exponential_ramp_to_value_at_time(10000.0, frames / SR)
        if name == "c2k":  # k-rate automation: the cutoff sweeps 100 Hz -> 8 kHz, one value per render quantum
            nq = (frames + RQ - 1) // RQ
            bq.frequency.set_block(0, np.geomspace(100.0, 8000.0, nq).astype(np.float32))
        node = node.connect(bq)
    if name in ("c2", "c2k"):
        node = node.connect(ctx.create_gain(gain=0.5))
    if name in ("t1", "c3", "c4"):
        from graphs import garage_ir  # the reference's parking-garage response, decoded + resampled to 48 kHz
        node = node.connect(ctx.create_convolver(buffer=waa.AudioBuffer(garage_ir(binding), SR)))
    if name == "c4":
        node = node.connect(ctx.create_stereo_panner(pan=0.1))
        node = node.connect(ctx.create_analyser(fft_size=2048, smoothing_time_constant=0.8))
    if name.startswith("iir"):  # IIRFilterNode, Butterworth low-pass of the given order
        from scipy import signal
        b, a = signal.butter(int(name[3:]), 0.25)
        node = node.connect(ctx.create_iir_filter(b, a))
    if name == "echo":
        node.connect(ctx.destination())
        node = node.connect(ctx.create_delay(1.0, delay_time=0.25)).connect(ctx.create_gain(gain=0.5))
    if name in ("fb", "fbq"):  # feedback echo, DelayNode (0.25 s) <-> Gain(0.5) [-> Biquad]
        delay = ctx.create_delay(1.0, delay_time=0.25)
        src.connect(delay)
        tail = delay
        if name == "fbq":
            tail = delay.connect(ctx.create_biquad_filter(type_="lowpass", frequency=4000.0))
        tail.connect(ctx.create_gain(gain=0.5)).connect(delay)
        tail.connect(ctx.destination())
    if name in ("os2", "os4"):  # WaveShaper with 2x / 4x oversampling (waveshaper.rs:409-481)
        node = node.connect(ctx.create_wave_shaper(curve=np.tanh(np.linspace(-3.0, 3.0, 2049)).astype(np.float32),
                                                   oversample="2x" if name == "os2" else "4x"))
    if name == "hrtf":  # PannerNode, HRTF panning model (panner.rs:781-829), static geometry
        waa.set_hrtf_database(os.path.join(ROOT, "tests", "golden", "IRC_1003_C.bin"))
        node = node.connect(ctx.create_panner(panning_model="HRTF", position=(1.0, 0.5, -0.5)))
    if name == "c5":
        src.playback_rate.set_value(1.5)
        src.set_loop(True)
        i = np.arange(2048, dtype=np.float32)
        curve = np.cos(np.float32(np.pi) + i * np.float32(np.pi) / np.float32(2047)).astype(np.float32)
        node = node.connect(ctx.create_wave_shaper(curve=curve))
    node.connect(ctx.destination())
    src.start()
    return ctx, src


# algorithmic bytes per context-quantum
ALG_BYTES = {"c2": 2048.0, "c2k": 2048.0, "c5": 2560.0, "c3": 362848.0, "t1": 362848.0 + 2048.0, "c4": 362848.0 + 2048.0 + 512.0}
ALG_BYTES["c1a"] = 2048.0
ALG_BYTES["fb"] = ALG_BYTES["fbq"] = 2048.0
ALG_BYTES["fm"] = 1024.0
ALG_BYTES["osc"] = 1024.0   # no input; 2 output channels x 128 frames x 4 B
ALG_BYTES["echo"] = 2048.0
ALG_BYTES["os2"] = ALG_BYTES["os4"] = 2048.0
ALG_BYTES["hrtf"] = 2048.0
# f32 FMA work per context-quantum of the compute-bound workloads (2 flops per multiply-add): both resampling stages of
# the oversampled WaveShaper as matrix products (2 channels x (128R x 256 + 128 x 256R)), the HRTF FIR (128 frames x 2 ears
# x 512 taps at 44.1 / 48 kHz -> 415 taps at 48 kHz)
ALG_FLOPS = {"os2": 2 * 2.0 * (256 * 256 + 128 * 512), "os4": 2 * 2.0 * (512 * 256 + 128 * 1024), "hrtf": 2.0 * 128 * 2 * 415}
IIR_ORDERS = (2, 4, 8, 12, 19)
for _o in IIR_ORDERS:
    ALG_BYTES[f"iir{_o}"] = 2048.0
DESCR = {
    "c2k": "C2 with k-rate automation: {n} contexts x {s:g} s, Biquad cutoff swept per render quantum ->Gain(0.5)->destination",
    "c2": "C2: {n} OfflineAudioContexts x {s:g} s @48kHz stereo, BufferSource->Biquad(lowpass 200Hz,Q1)->Gain(0.5)->destination",
    "t1": "T1: {n} contexts x {s:g} s, BufferSource->Biquad->Convolver(2ch x 178899-frame IR, 175 partitions)->destination",
    "c3": "C3: {n} contexts x {s:g} s, BufferSource->Convolver(2ch x 178899-frame IR)->destination",
    "c4": "C4: {n} contexts x {s:g} s, BufferSource->Biquad->Convolver->StereoPanner->Analyser->destination",
    "c5": "C5: {n} contexts x {s:g} s, BufferSource(playbackRate 1.5, loop)->WaveShaper(2048-pt)->destination",
}
DESCR["c1a"] = ("C1 a-rate variant x {n}: {s:g} s, BufferSource->Biquad(lowpass, frequency exponential ramp 10 Hz->10 kHz, "
                "per-sample coefficients)->destination")
DESCR["os2"] = "WaveShaper 2x: {n} contexts x {s:g} s, BufferSource(stereo)->WaveShaper(tanh 2049-pt, oversample 2x)->destination"
DESCR["os4"] = "WaveShaper 4x: {n} contexts x {s:g} s, BufferSource(stereo)->WaveShaper(tanh 2049-pt, oversample 4x)->destination"
DESCR["hrtf"] = "HRTF panner: {n} contexts x {s:g} s, BufferSource(stereo)->PannerNode(HRTF, static position)->destination"
DESCR["fm"] = "two-operator FM: {n} contexts x {s:g} s, Oscillator->Gain(300)->carrier.frequency, carrier->Gain->destination"
DESCR["osc"] = "subtractive voice: {n} contexts x {s:g} s, Oscillator(sawtooth 110 Hz, detuned)->Biquad(lowpass)->Gain->destination"
DESCR["echo"] = "feed-forward echo: {n} contexts x {s:g} s, BufferSource->destination + BufferSource->Delay(0.25s)->Gain(0.5)->destination"
DESCR["fb"] = "feedback echo: {n} contexts x {s:g} s, BufferSource->[Delay(0.25s)<->Gain(0.5)]->destination (+dry)"
DESCR["fbq"] = "filtered feedback echo: {n} contexts x {s:g} s, BufferSource->[Delay(0.25s)->Biquad->Gain(0.5)->back]->destination (+dry)"
for _o in IIR_ORDERS:
    DESCR[f"iir{_o}"] = "IIR: {n} contexts x {s:g} s, BufferSource->IIRFilter(Butterworth order %d)->destination" % _o


def cpu_baseline(waa, name, frames, target_wall=6.0):
    """The oracle ("port": a C restatement of the reference algorithm, NOT the Rust reference itself — no cargo on
    this box, probed) timed on this box's host cores, one context per thread on all threads, on a BOUNDED sample of
    the same workload: one batch of `cores` contexts is rendered repeatedly until the timed region lasts at least
    `target_wall` seconds (thread start-up and first-touch costs are amortised; the first, untimed render warms the
    pages).  Also reports the single-thread figure (one context on one thread), BASELINE.md section 3."""
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    orc = waa.bind(lib, "orc_")
    lib.orc_set_threads.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    lib.orc_rewind.argtypes = [ctypes.c_void_p]
    lib.orc_rewind.restype = ctypes.c_int32
    cores = os.cpu_count() or 1
    from graphs import white_noise

    def timed(n, fr, threads, wall_target):
        noise = white_noise(n, 2, fr)
        ctx, src = build_workload(waa, orc, name, n, fr, -1, None)
        if hasattr(src, "set_buffer_batch"):  # (the oscillator workloads have no input buffer)
            src.set_buffer_batch(noise, SR)
        ctx.prepare()
        lib.orc_set_threads(ctx._handle, threads)
        orc.check(orc.render(ctx._handle))  # untimed: the reference render; pages touched
        reps, wall = 0, 0.0
        while wall < wall_target and reps < 10000:
            orc.check(lib.orc_rewind(ctx._handle))  # (timing aid of the oracle: pre-render state again, untimed)
            t0 = time.perf_counter()
            orc.check(orc.render(ctx._handle))
            wall += time.perf_counter() - t0
            reps += 1
        ctx.close()
        return reps, wall

    # sample length: the full render for cheap graphs, shortened for the convolver graphs (~0.2 s of CPU per
    # context-second) so that one repetition stays well under the target
    conv = name in ("t1", "c3", "c4")
    sample_frames = min(frames, (RQ * 375 * 2) if conv else frames)  # 2 s of audio for convolver graphs
    sample_frames = (sample_frames // RQ) * RQ
    per_thread = 2 if not conv else 1
    n = cores * per_thread
    reps, wall = timed(n, sample_frames, cores, target_wall)
    nq = sample_frames // RQ
    out = {"value": n * nq * reps / wall, "unit": "quanta/s", "cores": cores, "kind": "port",
           "sample": f"{reps} x ({n} contexts x {sample_frames / SR:.2f} s of the same graph), one context per thread on "
                     f"{cores} threads, {wall:.2f} s of timed renders (the batch is rewound, untimed, between repetitions)",
           "rtf": n * (sample_frames / SR) * reps / wall,
           "reference_toolchain": "cargo/rustc probed on the GPU box: absent (no network): the Rust crate cannot be timed"}
    reps1, wall1 = timed(1, sample_frames, 1, min(2.0, target_wall))
    out["single_thread_rtf"] = (sample_frames / SR) * reps1 / wall1
    out["single_thread_quanta_per_s"] = nq * reps1 / wall1
    return out


def pmc_record(name, n_inst, frames):
    """rocprofv3 --pmc record of this workload (profiles/pmc_traffic.json; FETCH_SIZE and WRITE_SIZE are collected in
    separate passes of this same command and corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE x 2 on gfx950;
    the file records the numbers and their provenance).  None when no PMC data matches the configuration."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(path)).get(name)
        if rec and rec["contexts"] == n_inst and rec["frames"] == frames:
            return rec
    except (OSError, ValueError, KeyError):
        pass
    return None


DEFAULT_INSTANCES = {"c2": 1024, "c2k": 1024, "c1a": 1024, "t1": 1024, "c3": 512, "c4": 512, "c5": 2048}
F64_WORKLOADS = ("c2", "c2k", "c1a", "t1", "c4", "fbq", "osc")


def measure(torch, waa, hip, name, n_inst, seconds, steps, warmup, rank, world, local_rank, dist, backend):
    """One workload on this rank's GPU: build (untimed), first render = plan (timed separately as plan_ms), then the
    bench protocol (W untimed + K timed steps, barrier + sync on both sides, MAX over ranks).  Returns the record
    fields that depend on the workload (rank 0 assembles the line)."""
    frames = int(round(seconds * SR))
    nq = (frames + RQ - 1) // RQ
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0xA0D10 + rank)
    noise = torch.empty((n_inst, 2, frames), dtype=torch.float32, device="cuda").uniform_(-1.0, 1.0, generator=gen)
    ctx, _ = build_workload(waa, hip, name, n_inst, frames, local_rank, noise.data_ptr())
    ctx.prepare()

    from web_audio_api_rs_amd.sharding import timed_steps

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.render_async()  # the first launch builds the plan (graph planning, source scheduling replay, coefficient /
    ctx.sync()          # automation evaluation and their uploads) — outside the timed steps, reported as plan_ms
    first_ms = (time.perf_counter() - t0) * 1e3
    ctx.profile(True)
    ctx.profile_reset()
    elapsed = timed_steps(ctx.render_async, torch.cuda.synchronize, steps, warmup, dist=dist,
                          device_tensor=lambda v: torch.tensor([v], dtype=torch.float64,
                                                               device="cuda" if backend == "nccl" else "cpu"))
    ctx.sync()
    prof = sorted(ctx.profile_entries(), key=lambda e: -e[2])
    ctx.close()
    del noise
    torch.cuda.empty_cache()

    ms_per_step = elapsed / steps * 1e3
    total_launch_steps = steps + warmup  # the warm-up launches were event-timed too: normalise per launch
    kernel_ms = {n_: (ms / max(l, 1)) for n_, l, ms in prof}
    launches_per_step = {n_: l / total_launch_steps for n_, l, ms in prof}
    kernel_ms_per_step = sum(ms for _, _, ms in prof) / total_launch_steps
    dom = prof[0] if prof else ("none", 1, float("nan"))
    alg_bytes_step = ALG_BYTES[name] * n_inst * nq
    pmc = pmc_record(name, n_inst, frames)
    single = len(prof) == 1 and max(launches_per_step.values(), default=1) <= 1.5
    roof = {"bound": "hbm", "peak": 8000.0, "unit": "GB/s", "kernel_ms": kernel_ms, "launches_per_step": launches_per_step}
    if single:
        # one streaming kernel: SURVEY section 8(d)'s algorithmic bytes of one launch / its mean duration (HIP events)
        roof["kernel"] = dom[0]
        roof["achieved"] = alg_bytes_step / (dom[2] / max(dom[1], 1) * 1e-3) / 1e9
        roof["traffic"] = (pmc.get("bytes_per_launch") or pmc.get("bytes_per_step")) if pmc else None
        roof["algorithmic_bytes_per_launch"] = alg_bytes_step
    else:
        # several kernels (convolver pipelines, split chains, loops): the node-major design does not move the
        # reference's frequency-domain-delay-line bytes, so dividing THAT figure by the time says nothing (it exceeds
        # the peak).  achieved = HBM bytes the kernels really moved (PMC, per step) / the sum of their durations;
        # without a PMC record: the compulsory bytes (every input read once, every output written once).
        compulsory = 2048.0 * n_inst * nq if name not in ("c5",) else ALG_BYTES[name] * n_inst * nq
        traffic = pmc["bytes_per_step"] if pmc and "bytes_per_step" in pmc else None
        roof["kernel"] = "render (all kernels)"
        roof["traffic"] = traffic
        roof["achieved"] = (traffic if traffic else compulsory) / (kernel_ms_per_step * 1e-3) / 1e9
        roof["achieved_basis"] = "measured HBM traffic (rocprofv3 PMC, profiles/pmc_traffic.json)" if traffic else \
            "compulsory bytes (inputs read once + outputs written once): no PMC record for this configuration"
        roof["compulsory_bytes_per_step"] = compulsory
        roof["compulsory_frac"] = compulsory / (kernel_ms_per_step * 1e-3) / 1e9 / 8000.0
        if name in ("t1", "c3", "c4"):
            roof["reference_fdl_bytes"] = alg_bytes_step  # what the REFERENCE algorithm would move
        if pmc and "kernels" in pmc:
            roof["traffic_per_kernel"] = pmc["kernels"]
    roof["frac"] = roof["achieved"] / 8000.0
    if name in ALG_FLOPS:
        # compute-bound rows (f32 FMA; no MFMA format with enough mantissa except the f32 one, same peak): flops of the
        # dominant kernels / their time against the 157.3 TFLOP/s f32 peak of MI355X_MICROARCH.md
        flops = ALG_FLOPS[name] * n_inst * nq
        comp_ms = sum(ms / max(l, 1) * (l / total_launch_steps) for n_, l, ms in prof if n_.startswith(("qgemm", "hrtf")))
        roof.update({"bound": "valu_f32", "peak": 157.3, "unit": "TFLOP/s", "achieved": flops / (comp_ms * 1e-3) / 1e12,
                     "algorithmic_flops_per_step": flops, "compute_kernel_ms_per_step": comp_ms})
        roof["frac"] = roof["achieved"] / 157.3
        if name in ("os2", "os4") and not any(os.environ.get(k) for k in ("WAA_QGEMM_FMA", "WAA_QGEMM_F32")):
            # the resampling products run on the bf16 matrix cores as SIX bf16 products per f32 product (exact three-way
            # split of both operands, f32-grade result: DESIGN.md 3.5): the ceiling of that method is the dense bf16 MFMA
            # peak / 6, in f32-equivalent flops; `mfma_flops_per_step` is what the matrix cores really execute
            roof.update({"bound": "mfma", "peak": 2500.0 / 6.0, "peak_basis": "2.5 PFLOP/s dense bf16 MFMA / 6 products per f32 product",
                         "mfma_flops_per_step": 6.0 * flops, "vs_f32_vector_peak": roof["achieved"] / 157.3})
            roof["frac"] = roof["achieved"] / (2500.0 / 6.0)
    return {
        "value": world * n_inst * nq * steps / elapsed,
        "ms_per_step": ms_per_step,
        "plan_ms": max(first_ms - kernel_ms_per_step, 0.0),
        "dtype": "f64" if name in F64_WORKLOADS or name.startswith("iir") else "f32",
        "config": {"workload": DESCR[name].format(n=n_inst, s=seconds), "contexts_per_gpu": n_inst,
                   "sample_rate": SR, "render_seconds": seconds, "quanta_per_context": nq,
                   "parallelism": f"{world} independent batch(es), no collective"},
        "real_time_factor": world * n_inst * seconds * steps / elapsed,
        "roofline": roof,
    }


def e2e_record(torch, waa, hip, n_inst, seconds, local_rank, n_sub=8):
    """What the drop-in boundary costs when it is handed HOST buffers (never `value`): host noise ->
    waa_source_set_buffer_batch -> waa_render -> waa_download_all for the C2 graph, (a) as one batch, (b) as n_sub
    sub-batches driven by n_sub host threads (ctypes releases the GIL; every batch has its own stream and its transfers run
    on it), pipelined so that the upload of one sub-batch overlaps the download of another.  Host buffers are pinned (torch)."""
    import threading
    frames = int(round(seconds * SR))
    host_in = torch.empty((n_inst, 2, frames), dtype=torch.float32, pin_memory=True).uniform_(-1.0, 1.0)
    host_out = torch.empty((n_inst, 2, frames), dtype=torch.float32, pin_memory=True)
    import ctypes as C
    FP = C.POINTER(C.c_float)

    # one upload and one download at a time: sub-batch i + 1 uploads while sub-batch i downloads (PCIe is full duplex:
    # tools/pcie_probe.py, 57 GB/s each way, 97 GB/s both at once).  Without the two locks all uploads run side by side,
    # finish together, and all downloads follow: a sum again, not a pipeline.
    class Turn:  # sub-batches take the link in index order
        def __init__(self):
            self.cv, self.next = threading.Condition(), 0

        def wait(self, k):
            with self.cv:
                self.cv.wait_for(lambda: self.next == k)

        def done(self):
            with self.cv:
                self.next += 1
                self.cv.notify_all()

    turns = {}

    def run(lo, hi, k=0):
        up, down = turns["up"], turns["down"]
        ctx, src = build_workload(waa, hip, "c2", hi - lo, frames, local_rank, None)
        ctx.prepare()
        sl = host_in[lo:hi]
        up.wait(k)
        hip.check(hip.source_set_buffer_batch(ctx._handle, src.id, C.cast(sl.data_ptr(), FP), 2, frames, SR))
        up.done()
        hip.check(hip.render(ctx._handle))
        hip.check(hip.sync(ctx._handle))
        down.wait(k)
        hip.check(hip.download_all(ctx._handle, C.cast(host_out[lo:hi].data_ptr(), FP)))
        down.done()
        ctx.close()

    def timed(parts):
        bounds = [(k * n_inst // parts, (k + 1) * n_inst // parts, k) for k in range(parts)]
        turns["up"], turns["down"] = Turn(), Turn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ths = [threading.Thread(target=run, args=b) for b in bounds]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        return (time.perf_counter() - t0) * 1e3

    timed(n_sub)  # warm-up: allocator, page registration
    one = min(timed(1) for _ in range(2))
    split = min(timed(n_sub) for _ in range(2))
    nbytes = 2.0 * n_inst * 2 * frames * 4
    rec = {"workload": "c2", "contexts": n_inst, "single_batch_ms": one, f"split_{n_sub}_batches_ms": split,
           "host_bytes_moved": nbytes, "effective_GBps_single": nbytes / one / 1e6,
           "effective_GBps_split": nbytes / split / 1e6,
           "note": "host (pinned) -> set_buffer_batch -> render -> download_all, batch creation and planning included; "
                   "PCIe-bound, reported for the boundary only; the split run pipelines its sub-batches (upload of one "
                   "while another downloads: the link is full duplex, tools/pcie_probe.py)"}
    # the same with the input handed over as decoded 16-bit PCM (waa_source_set_buffer_pcm16_batch: half the upload,
    # sample conversion on the device) — what a caller that holds WAV data would do
    try:
        pcm = torch.empty((n_inst, frames, 2), dtype=torch.int16, pin_memory=True).random_(-32768, 32767)
        I16 = C.POINTER(C.c_int16)

        def run_pcm(lo, hi, k=0):
            up, down = turns["up"], turns["down"]
            ctx, src = build_workload(waa, hip, "c2", hi - lo, frames, local_rank, None)
            ctx.prepare()
            up.wait(k)
            hip.check(hip.source_set_buffer_pcm16_batch(ctx._handle, src.id, C.cast(pcm[lo:hi].data_ptr(), I16), 2, frames, SR))
            up.done()
            hip.check(hip.render(ctx._handle))
            hip.check(hip.sync(ctx._handle))
            down.wait(k)
            hip.check(hip.download_all(ctx._handle, C.cast(host_out[lo:hi].data_ptr(), FP)))
            down.done()
            ctx.close()

        def timed_pcm(parts):
            bounds = [(k * n_inst // parts, (k + 1) * n_inst // parts, k) for k in range(parts)]
            turns["up"], turns["down"] = Turn(), Turn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ths = [threading.Thread(target=run_pcm, args=b_) for b_ in bounds]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            return (time.perf_counter() - t0) * 1e3

        timed_pcm(1)
        rec["pcm16_single_batch_ms"] = min(timed_pcm(1) for _ in range(2))
        timed_pcm(n_sub)
        rec[f"pcm16_split_{n_sub}_batches_ms"] = min(timed_pcm(n_sub) for _ in range(2))
        rec["pcm16_host_bytes_moved"] = n_inst * 2 * frames * (2 + 4.0)
    except Exception as e:  # reporting only
        rec["pcm16_error"] = repr(e)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=sorted(ALG_BYTES))
    ap.add_argument("--instances", type=int, default=None, help="contexts per GPU")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the nested T1/C3/C5/C1a records and the e2e record")
    args = ap.parse_args()

    import torch
    import web_audio_api_rs_amd as waa

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    # one rank per GPU.  WAA_BENCH_BACKEND=gloo + WAA_BENCH_SHARE_GPU=1 let the N > 1 plumbing be exercised on a
    # single-GPU box (ranks share device 0; RCCL refuses two ranks on one device) — never used for reported numbers.
    backend = os.environ.get("WAA_BENCH_BACKEND", "nccl")
    if os.environ.get("WAA_BENCH_SHARE_GPU") == "1":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    name = args.workload
    n_inst = args.instances or DEFAULT_INSTANCES.get(name, 1024)
    hip = waa.default_binding()
    rec = measure(torch, waa, hip, name, n_inst, args.seconds, args.steps, args.warmup, rank, world, local_rank, dist,
                  backend)
    # The north-star target graph (T1) and the other BASELINE configs ride along in the same line, so that one driver
    # run verifies them all: every rank renders them (same barrier protocol), rank 0 reports.  Fewer steps each.
    extra = {}
    default_run = name == "c2" and args.instances is None and args.seconds == 10.0 and not args.no_extra
    if default_run:
        for sub in ("t1", "c3", "c4", "c5", "c1a", "os2", "hrtf", "echo"):
            try:
                extra[sub] = measure(torch, waa, hip, sub, DEFAULT_INSTANCES.get(sub, 1024), args.seconds, max(3, args.steps // 2),
                                     min(args.warmup, 2) or 1, rank, world, local_rank, dist, backend)
                extra[sub]["steps"] = max(3, args.steps // 2)
            except Exception as e:  # a sub-record never takes the headline line down
                extra[sub] = {"error": repr(e)}

    if rank == 0:
        out = {
            "metric": "render quanta/sec (48kHz, 128-frame)",
            "value": rec["value"],
            "unit": "quanta/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": rec["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": rec["dtype"],
            "data": "synthetic",
            "config": rec["config"],
            "real_time_factor": rec["real_time_factor"],
            "plan_ms": rec["plan_ms"],
            "roofline": rec["roofline"],
        }
        if extra:
            out["workloads"] = extra
        if world > 1:
            out["cpu_baseline"] = None  # timed at N = 1 only: the host cores are shared by the ranks
        elif not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(waa, name, int(round(args.seconds * SR)))
                if out["cpu_baseline"]:
                    out["gpu_over_cpu_rtf"] = out["real_time_factor"] / out["cpu_baseline"]["rtf"]
                if default_run and "t1" in extra and "error" not in extra["t1"]:
                    cb = cpu_baseline(waa, "t1", int(round(args.seconds * SR)), target_wall=5.0)
                    extra["t1"]["cpu_baseline"] = cb
                    if cb:
                        extra["t1"]["gpu_over_cpu_rtf"] = extra["t1"]["real_time_factor"] / cb["rtf"]
            except Exception as e:  # the baseline is reporting only; never fail the bench line on it
                out["cpu_baseline"] = {"error": repr(e)}
        if default_run and world == 1:
            try:
                out["e2e"] = e2e_record(torch, waa, hip, n_inst, args.seconds, local_rank)
            except Exception as e:
                out["e2e"] = {"error": repr(e)}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
