"""tools/placement_fill_probe.py — is the per-allocation spread of C2 visible to a plain fill of
the batch's output buffer?  Per trial: the kernel's HIP-event time on a freshly created batch, and the time of hipMemset over
its 3.9 GB output.  If the two correlate, the library can pick its output allocation by probing candidates.  (GPU box)"""
import ctypes
import os
os.environ.setdefault("WAA_USE_MEASURE_LIB", "1")  # A/B and probe tools flip measurement switches: libwaa_hip_measure.so
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
import web_audio_api_rs_amd as waa  # noqa: E402

hiprt = ctypes.CDLL("libamdhip64.so")
hiprt.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
hiprt.hipDeviceSynchronize.argtypes = []


def fill_ms(ptr, nbytes, reps=6):
    best = 1e9
    for _ in range(reps):
        hiprt.hipDeviceSynchronize()
        t0 = time.perf_counter()
        hiprt.hipMemset(ptr, 0, nbytes)
        hiprt.hipDeviceSynchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best


def timed(ctx, iters=10):
    ctx.profile_reset()
    for _ in range(iters):
        ctx.render_async()
    ctx.sync()
    return {n: round(ms / iters, 3) for n, l, ms in ctx.profile_entries()}


n_inst, frames = 1024, 480000
hip = waa.default_binding()
noise = torch.empty((n_inst, 2, frames), dtype=torch.float32, device="cuda").uniform_(-1, 1)
spacers = []
for trial in range(int(os.environ.get("TRIALS", "12"))):
    if trial % 2 == 1:
        spacers.append(torch.empty(((trial * 37 + 11) << 20,), dtype=torch.uint8, device="cuda"))
    ctx, _ = bench.build_workload(waa, hip, "c2", n_inst, frames, 0, noise.data_ptr())
    ctx.prepare()
    ctx.render_async()
    ctx.sync()
    ctx.profile(True)
    k = timed(ctx)
    out = ctx.output_device()
    ptr, inst_stride = out[0], out[1]
    nbytes = n_inst * inst_stride * 4
    f = fill_ms(ptr, nbytes)
    k2 = timed(ctx)
    print(f"trial {trial} out {ptr:#x} kernel {k} fill {f:.3f} ms ({nbytes / f / 1e6:.0f} GB/s) kernel again {k2}", flush=True)
    ctx.close()
