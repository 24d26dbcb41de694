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
    if name in ("c2", "c2k", "t1", "c4"):
        bq = ctx.create_biquad_filter(type_="lowpass", frequency=200.0, q=1.0)
        if name == "c2k":  # k-rate automation: the cutoff sweeps 100 Hz -> 8 kHz, one value per render quantum
            nq = (frames + RQ - 1) // RQ
            bq.frequency.set_block(0, np.geomspace(100.0, 8000.0, nq).astype(np.float32))
        node = node.connect(bq)
    if name in ("c2", "c2k"):
        node = node.connect(ctx.create_gain(gain=0.5))
    if name in ("t1", "c3", "c4"):
        from graphs import garage_like_ir
        node = node.connect(ctx.create_convolver(buffer=waa.AudioBuffer(garage_like_ir(), SR)))
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
ALG_BYTES["fb"] = ALG_BYTES["fbq"] = 2048.0
ALG_BYTES["fm"] = 1024.0
ALG_BYTES["osc"] = 1024.0   # no input; 2 output channels x 128 frames x 4 B
ALG_BYTES["echo"] = 2048.0
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
DESCR["fm"] = "two-operator FM: {n} contexts x {s:g} s, Oscillator->Gain(300)->carrier.frequency, carrier->Gain->destination"
DESCR["osc"] = "subtractive voice: {n} contexts x {s:g} s, Oscillator(sawtooth 110 Hz, detuned)->Biquad(lowpass)->Gain->destination"
DESCR["echo"] = "feed-forward echo: {n} contexts x {s:g} s, BufferSource->destination + BufferSource->Delay(0.25s)->Gain(0.5)->destination"
DESCR["fb"] = "feedback echo: {n} contexts x {s:g} s, BufferSource->[Delay(0.25s)<->Gain(0.5)]->destination (+dry)"
DESCR["fbq"] = "filtered feedback echo: {n} contexts x {s:g} s, BufferSource->[Delay(0.25s)->Biquad->Gain(0.5)->back]->destination (+dry)"
for _o in IIR_ORDERS:
    DESCR[f"iir{_o}"] = "IIR: {n} contexts x {s:g} s, BufferSource->IIRFilter(Butterworth order %d)->destination" % _o


def cpu_baseline(waa, name, frames, target_wall=12.0):
    """The oracle ("port": a C restatement of the reference algorithm, NOT the Rust reference itself) timed
    on this box's host cores, one context per thread, on a BOUNDED sample of the same workload: a short
    single-thread calibration render fixes the sample duration so the timed run takes ~target_wall seconds."""
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    orc = waa.bind(lib, "orc_")
    lib.orc_set_threads.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    cores = os.cpu_count() or 1
    from graphs import white_noise

    def run(n, fr, threads):
        noise = white_noise(n, 2, fr)
        ctx, src = build_workload(waa, orc, name, n, fr, -1, None)
        if hasattr(src, "set_buffer_batch"):  # (the oscillator workload has no input buffer)
            src.set_buffer_batch(noise, SR)
        ctx.prepare()
        lib.orc_set_threads(ctx._handle, threads)
        t0 = time.perf_counter()
        orc.check(orc.render(ctx._handle))
        wall = time.perf_counter() - t0
        ctx.close()
        return wall

    # calibration under the same contention as the timed run: every thread renders one short context
    cal_frames = min(frames, 128 * 160)
    t_cal = max(run(cores, cal_frames, cores), 1e-4)
    sec_per_ctx_sec = t_cal / (cal_frames / SR)  # wall seconds per rendered second with all threads busy
    sample_frames = int(min(frames, max(cal_frames, (target_wall / sec_per_ctx_sec) * SR)))
    sample_frames = (sample_frames // RQ) * RQ
    per_thread = max(1, int(target_wall / max(sec_per_ctx_sec * sample_frames / SR, 1e-6)))
    per_thread = min(per_thread, 16, max(1, int(2e9 / (sample_frames * 8.0) / cores)))  # <= 2 GB of host noise
    n = cores * per_thread
    wall = run(n, sample_frames, cores)
    nq = sample_frames // RQ
    return {"value": n * nq / wall, "unit": "quanta/s", "cores": cores, "kind": "port",
            "sample": f"{n} contexts x {sample_frames / SR:.2f} s of the same graph, one context per thread on {cores} "
                      f"threads, wall {wall:.2f} s (all-thread calibration {t_cal:.2f} s for {cal_frames / SR:.2f} s)",
            "rtf": n * (sample_frames / SR) / wall}


def pmc_traffic(name, n_inst, frames):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE are collected in separate runs of this same command, corrected as MI355X_MICROARCH.md prescribes:
    profiles/pmc_traffic.json records the numbers and their provenance).  null when no PMC data matches."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(path)).get(name)
        if rec and rec["contexts"] == n_inst and rec["frames"] == frames:
            return rec["bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=sorted(ALG_BYTES))
    ap.add_argument("--instances", type=int, default=None, help="contexts per GPU")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    n_inst = args.instances or {"c2": 1024, "c2k": 1024, "t1": 1024, "c3": 512, "c4": 512, "c5": 2048}.get(name, 1024)
    frames = int(round(args.seconds * SR))
    nq = (frames + RQ - 1) // RQ
    hip = waa.default_binding()

    gen = torch.Generator(device="cuda")
    gen.manual_seed(0xA0D10 + rank)
    noise = torch.empty((n_inst, 2, frames), dtype=torch.float32, device="cuda").uniform_(-1.0, 1.0, generator=gen)
    ctx, _ = build_workload(waa, hip, name, n_inst, frames, local_rank, noise.data_ptr())
    ctx.prepare()

    from web_audio_api_rs_amd.sharding import timed_steps

    ctx.render_async()  # first launch builds the plan (uploads schedules / coefficients) outside any timing
    ctx.sync()
    ctx.profile(True)
    ctx.profile_reset()
    elapsed = timed_steps(ctx.render_async, torch.cuda.synchronize, args.steps, args.warmup, dist=dist,
                          device_tensor=lambda v: torch.tensor([v], dtype=torch.float64,
                                                               device="cuda" if backend == "nccl" else "cpu"))
    ctx.sync()
    # the warmup launches were also event-timed: normalise per launch below

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        total_quanta = world * n_inst * nq
        value = total_quanta * args.steps / elapsed
        prof = sorted(ctx.profile_entries(), key=lambda e: -e[2])
        dom = prof[0] if prof else ("none", 1, float("nan"))
        total_launch_steps = args.steps + args.warmup
        kernel_ms = {n_: (ms / max(l, 1)) for n_, l, ms in prof}
        launches_per_step = {n_: l / total_launch_steps for n_, l, ms in prof}
        # roofline of the dominant kernel: algorithmic bytes of one launch / its mean duration
        alg_bytes_step = ALG_BYTES[name] * n_inst * nq
        dom_share = 1.0
        if name in ("c3", "t1", "c4") or len(prof) > 1 or max(launches_per_step.values(), default=1) > 1.5:
            # several kernels share the algorithmic bytes of the FDL: attribute them to the whole render
            achieved = alg_bytes_step / (sum(ms for _, _, ms in prof) / total_launch_steps * 1e-3) / 1e9
            dom_name = "render (all kernels)"
        else:
            achieved = alg_bytes_step * dom_share / (dom[2] / max(dom[1], 1) * 1e-3) / 1e9
            dom_name = dom[0]
        out = {
            "metric": "render quanta/sec (48kHz, 128-frame)",
            "value": value,
            "unit": "quanta/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if name in ("c2", "c2k", "t1", "c4", "fbq", "osc") or name.startswith("iir") else "f32",
            "data": "synthetic",
            "config": {"workload": DESCR[name].format(n=n_inst, s=args.seconds), "contexts_per_gpu": n_inst,
                       "sample_rate": SR, "render_seconds": args.seconds, "quanta_per_context": nq,
                       "parallelism": f"{world} independent batch(es), no collective"},
            "real_time_factor": world * n_inst * args.seconds * args.steps / elapsed,
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": pmc_traffic(name, n_inst, frames),
                         "algorithmic_bytes_per_launch": alg_bytes_step,
                         "kernel_ms": kernel_ms, "launches_per_step": launches_per_step},
        }
        if world > 1:
            out["cpu_baseline"] = None  # timed at N = 1 only: the host cores are shared by the ranks
        elif not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(waa, name, frames)
                if out["cpu_baseline"]:
                    out["gpu_over_cpu_rtf"] = out["real_time_factor"] / out["cpu_baseline"]["rtf"]
            except Exception as e:  # the baseline is reporting only; never fail the bench line on it
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
