"""tools/arena_probe.py — the graded arena against plain hipMalloc, same process (GPU box; measurement build).
1. C2 on plain hipMalloc, the first batch of the process (`cold`), product kernel / its copy-only form (WAA_STREAM_DEBUG=1);
2. waa_device_arena_reserve_graded: the units' grades;
3. C2 out of the arena: the caller's noise tensor as the source (read in place), then the noise uploaded into an arena payload;
4. echo and T1 out of the arena and (arena released) on plain hipMalloc."""
import os
os.environ.setdefault("WAA_USE_MEASURE_LIB", "1")
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import web_audio_api_rs_amd as waa  # noqa: E402


def timed(ctx, iters=8):
    ctx.profile(True)
    ctx.profile_reset()
    for _ in range(iters):
        ctx.render_async()
    ctx.sync()
    return {n: round(ms / max(l, 1), 4) for n, l, ms in ctx.profile_entries()}


def run(hip, name, noise, tag, n_inst=1024, frames=480000, debug=("", "1")):
    ctx, _ = bench.build_workload(waa, hip, name, n_inst, frames, 0, noise.data_ptr())
    ctx.prepare()
    ctx.render_async()
    ctx.sync()
    line = [f"{tag:34s} {name} out {ctx.output_device()[0]:#x}"]
    for dbg in debug:
        if dbg:
            os.environ["WAA_STREAM_DEBUG"] = dbg
        line.append(f"{'dbg' + dbg if dbg else 'product'}: {timed(ctx)}")
        os.environ.pop("WAA_STREAM_DEBUG", None)
    print("  ".join(line), flush=True)
    ctx.close()


def main():
    hip = waa.default_binding()
    n_inst, frames = 1024, 480000
    noise = torch.empty((n_inst, 2, frames), dtype=torch.float32, device="cuda").uniform_(-1, 1)
    print(f"noise (torch, the first allocation of the process) at {noise.data_ptr():#x}")
    for k in range(3):
        run(hip, "c2", noise, f"cold, batch {k}")
    run(hip, "echo", noise, "cold", debug=("",))
    run(hip, "t1", noise, "cold", debug=("",))
    arena_gb, cand_gb = int(os.environ.get("ARENA_GB", "64")), int(os.environ.get("CAND_GB", "200"))
    hip.check(hip.device_arena_reserve_graded(0, arena_gb << 30, cand_gb << 30))
    g = waa.arena_grades(hip, 0)
    print({k: v for k, v in g.items() if k != "unit_ms"})
    print("kept units (ms):", " ".join(f"{v:.3f}" for v in g["unit_ms"]))
    for k in range(2):
        run(hip, "c2", noise, f"arena, batch {k}")
    run(hip, "echo", noise, "arena", debug=("",))
    run(hip, "t1", noise, "arena", debug=("",))
    # the source out of the arena too: a second noise tensor cannot be placed by torch, so copy it into a batch-owned payload —
    # set_buffer_batch from a HOST array uploads into an arena piece (top end: read-only)
    host = np.random.default_rng(1).uniform(-1, 1, (n_inst, 2, frames)).astype(np.float32)
    ctx = waa.OfflineAudioContext(2, frames, 48000.0, n_instances=n_inst, binding=hip, device=0)
    src = ctx.create_buffer_source()
    src.set_buffer_batch(host, 48000.0)
    flt = ctx.create_biquad_filter(type_="lowpass", frequency=200.0, q=1.0)
    gain = ctx.create_gain(gain=0.5)
    src.connect(flt).connect(gain).connect(ctx.destination())
    src.start()
    ctx.prepare()
    ctx.render_async()
    ctx.sync()
    print(f"{'arena, source uploaded into it':34s} c2 out {ctx.output_device()[0]:#x}  product: {timed(ctx)}", flush=True)
    os.environ["WAA_STREAM_DEBUG"] = "1"
    print(f"{'':34s}   dbg1: {timed(ctx)}", flush=True)
    os.environ.pop("WAA_STREAM_DEBUG")
    ctx.close()
    print(waa.arena_stats(hip, 0))
    hip.check(hip.device_arena_reserve(0, 0))
    for k in range(2):
        run(hip, "c2", noise, f"after the arena, plain, batch {k}")


if __name__ == "__main__":
    main()
