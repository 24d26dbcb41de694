"""tools/slab_probe.py [arena_gb] [trials] — does a device arena reserved at process start remove the slow kind of C2 batch?
(the same biquad_stream_kernel runs at ~1.33 ms or ~1.55 ms depending on which hipMalloc served the
batch's 3.9 GB output buffer.)  arena_gb > 0: waa_device_arena_reserve(0, arena_gb GiB) is the FIRST device allocation of the
process — before torch creates its context's caches and the noise — and every batch's output is carved from it; 0: plain
hipMalloc per batch.  Each trial creates a fresh C2 batch (1024 contexts x 10 s), with spacer allocations kept alive in between
like tools/placement_probe.py, and prints the output pointer and the kernel's HIP-event mean over 10 renders.  Run both, e.g.
    python tools/slab_probe.py 0 10 > gpurun_out/slab_off.txt; python tools/slab_probe.py 5 10 > gpurun_out/slab_on.txt"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402  (imported first so that the library binds the HIP runtime torch ships — two runtimes in one process
#                           see no devices; importing does not touch the device yet)

import web_audio_api_rs_amd as waa  # noqa: E402

arena_gb = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 10
hip = waa.default_binding()
if arena_gb > 0:
    hip.check(hip.device_arena_reserve(0, int(arena_gb * (1 << 30))))  # the first device allocation of the process

import bench  # noqa: E402

n_inst, frames = 1024, 480000
noise = torch.empty((n_inst, 2, frames), dtype=torch.float32, device="cuda").uniform_(-1, 1)
spacers, times = [], []
for trial in range(trials):
    if trial % 2 == 1:
        spacers.append(torch.empty(((trial * 37 + 11) << 20,), dtype=torch.uint8, device="cuda"))
    ctx, _ = bench.build_workload(waa, hip, "c2", n_inst, frames, 0, noise.data_ptr())
    ctx.prepare()
    ctx.render_async()
    ctx.sync()
    ctx.profile(True)
    ctx.profile_reset()
    for _ in range(10):
        ctx.render_async()
    ctx.sync()
    ms = {n: round(t / 10, 4) for n, l, t in ctx.profile_entries()}
    times.append(ms.get("biquad_stream_kernel"))
    print(f"arena {arena_gb:g} GiB trial {trial} out {ctx.output_device()[0]:#x} {ms}", flush=True)
    ctx.close()
t = sorted(x for x in times if x)
print(f"arena {arena_gb:g} GiB: min {t[0]} median {t[len(t) // 2]} max {t[-1]} ms over {len(t)} fresh batches")
