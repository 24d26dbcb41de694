"""A/B of the two convolver FFT kernels in one process: python tools/ab_conv_fft.py [n_inst]
WAA_CONV_FFT_PLAIN=1 selects the one-FFT-per-workgroup kernel, otherwise N = 16384 uses the persistent pipelined
kernel.  Prints per-kernel means (ms per render) for T1 and checks that both kernels give bit-identical output."""
import os
os.environ.setdefault("WAA_USE_MEASURE_LIB", "1")  # A/B and probe tools flip measurement switches: libwaa_hip_measure.so
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import web_audio_api_rs_amd as waa  # noqa: E402

n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
frames = 480000
hip = waa.default_binding()
noise = torch.empty((n_inst, 2, frames), dtype=torch.float32, device="cuda").uniform_(-1, 1)

def render_small(mode, name, n_small=6):
    os.environ.pop("WAA_CONV_FFT_PLAIN", None)
    if mode == "plain":
        os.environ["WAA_CONV_FFT_PLAIN"] = "1"
    small = noise[:n_small].contiguous()
    ctx, _ = bench.build_workload(waa, hip, name, n_small, frames, 0, small.data_ptr())
    out = ctx.start_rendering_sync().data
    ctx.close()
    return out


for name in ("t1",):
    a, b = render_small("plain", name), render_small("pipe", name)
    print(name, "plain vs pipelined bit-identical:", bool(np.array_equal(a, b)), "max |d|", float(np.abs(a - b).max()), flush=True)
for name in ("t1", "c3"):
    for mode in ("plain", "pipe"):
        os.environ.pop("WAA_CONV_FFT_PLAIN", None)
        if mode == "plain":
            os.environ["WAA_CONV_FFT_PLAIN"] = "1"
        ctx, _ = bench.build_workload(waa, hip, name, n_inst, frames, 0, noise.data_ptr())
        ctx.prepare()
        ctx.render_async()
        ctx.sync()
        ctx.profile(True)
        ctx.profile_reset()
        for _ in range(3):
            ctx.render_async()
        ctx.sync()
        print(name, mode, {n: round(ms / 3, 3) for n, l, ms in ctx.profile_entries()}, flush=True)
        ctx.close()
