"""Per-launch kernel time of a bench.py workload from a cold start: python tools/warmup_probe.py [workload] [launches]
(HIP-event duration of every render in order; shows how many launches the GPU needs to reach its steady clocks —
the reason bench.py's default warm-up is longer than two steps)."""
import os
os.environ.setdefault("WAA_USE_MEASURE_LIB", "1")  # A/B and probe tools flip measurement switches: libwaa_hip_measure.so
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
import web_audio_api_rs_amd as waa  # noqa: E402


def main(argv):
    name = argv[0] if argv else "c2"
    n = int(argv[1]) if len(argv) > 1 else 60
    n_inst, frames = bench.DEFAULT_INSTANCES.get(name, 1024), 480000
    hip = waa.default_binding()
    noise = torch.empty((n_inst, 2, frames), dtype=torch.float32, device="cuda").uniform_(-1, 1)
    ctx, _ = bench.build_workload(waa, hip, name, n_inst, frames, 0, noise.data_ptr())
    ctx.prepare()
    ctx.render_async()
    ctx.sync()
    time.sleep(float(os.environ.get("PROBE_IDLE", "0.5")))  # let the device fall idle, like a fresh process would find it
    ctx.profile(True)
    out = []
    for _ in range(n):
        ctx.profile_reset()
        ctx.render_async()
        ctx.sync()
        out.append(round(sum(ms for _, _, ms in ctx.profile_entries()), 3))
    print(name, "per-launch ms (synchronised after every launch):", out)
    ctx.profile_reset()
    for _ in range(n):
        ctx.render_async()
    ctx.sync()
    print(name, "mean of", n, "back-to-back launches:", round(sum(ms for _, _, ms in ctx.profile_entries()) / n, 3))
    ctx.close()


if __name__ == "__main__":
    main(sys.argv[1:])
