"""tools/plan_time_probe.py [workload ...] — host planning time of bench.py's workloads on a PLAN-ONLY batch (no device: runs in
the authoring container): what build_plan costs on the host, per workload, best of 5.  The first render of a fresh batch pays
this once (bench.py's `one_shot` / first_render_ms)."""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import bench  # noqa: E402
import web_audio_api_rs_amd as waa  # noqa: E402

hip = waa.default_binding()
names = sys.argv[1:] or ["c2", "t1", "c3", "c4", "c5", "c1a", "os2", "os4", "hrtf", "echo", "fb", "fbq", "fm", "osc", "trem"]
frames = 480000
for name in names:
    n_inst = bench.DEFAULT_INSTANCES.get(name, 1024)
    best, head = 1e9, ""
    for rep in range(5):
        ctx, src = bench.build_workload(waa, hip, name, n_inst, frames, waa.PLAN_ONLY, None)
        if hasattr(src, "set_buffer") and name not in ("fm", "osc"):
            src.set_buffer(waa.AudioBuffer(np.zeros((2, 65536 if name == "c5" else frames), np.float32), 48000.0))
        ctx.prepare()
        t0 = time.perf_counter()
        text = ctx.plan_describe()
        ms = (time.perf_counter() - t0) * 1e3
        if ms < best:
            best, head = ms, text.splitlines()[0].split("| timing: ")[-1]
        ctx.close()
    print(f"{name:5s} {n_inst:5d} ctx  plan {best:8.2f} ms   {head}", flush=True)
