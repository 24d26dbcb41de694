"""tools/timecourse_probe.py [seconds] — kernel time of one C2 batch rendered back to back for a while: per-block means
(100 renders per block) against the time since the first render.  Shows whether the device changes its operating point
(clocks / power state) under sustained load.  (GPU box)"""
import os
os.environ.setdefault("WAA_USE_MEASURE_LIB", "1")  # A/B and probe tools flip measurement switches: libwaa_hip_measure.so
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
import web_audio_api_rs_amd as waa  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
n_inst, frames = 1024, 480000
hip = waa.default_binding()
noise = torch.empty((n_inst, 2, frames), dtype=torch.float32, device="cuda").uniform_(-1, 1)
ctx, _ = bench.build_workload(waa, hip, "c2", n_inst, frames, 0, noise.data_ptr())
ctx.prepare()
ctx.render_async()
ctx.sync()
ctx.profile(True)
t0 = time.time()
out = []
while time.time() - t0 < seconds:
    ctx.profile_reset()
    for _ in range(100):
        ctx.render_async()
    ctx.sync()
    out.append((round(time.time() - t0, 2), round([ms for n, l, ms in ctx.profile_entries()][0] / 100, 3)))
print(out)
