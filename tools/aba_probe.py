"""tools/aba_probe.py — C2 kernel time against the number of contexts, several freshly created batches each: 1024 contexts
are exactly the 2048 wavefronts the device holds at once (8 per CU), fewer leave slack.  Separates "the memory system is
slower for this batch" from "a few workgroups did not get a slot in the first round".  (GPU box)"""
import os
os.environ.setdefault("WAA_USE_MEASURE_LIB", "1")  # A/B and probe tools flip measurement switches: libwaa_hip_measure.so
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
import web_audio_api_rs_amd as waa  # noqa: E402
from placement_probe import timed  # noqa: E402

frames = 480000
hip = waa.default_binding()
noise = torch.empty((1032, 2, frames), dtype=torch.float32, device="cuda").uniform_(-1, 1)
for n_inst in [int(x) for x in os.environ.get("COUNTS", "960,1000,1016,1024,1032").split(",")]:
    ts = []
    for rep in range(int(os.environ.get("REPS", "6"))):
        ctx, _ = bench.build_workload(waa, hip, "c2", n_inst, frames, 0, noise.data_ptr())
        ctx.prepare()
        ctx.render_async()
        ctx.sync()
        ctx.profile(True)
        ts.append(list(timed(ctx).values())[0])
        ctx.close()
    print(n_inst, "contexts:", ts, "us per context (min, max): %.3f %.3f" % (min(ts) / n_inst * 1e3, max(ts) / n_inst * 1e3), flush=True)
