"""tools/placement_probe.py [workload] — does the kernel time depend on WHERE the buffers are?  Renders one bench.py workload
from freshly created batches (spacer allocations of varying size in between) and prints the output pointer and the
per-kernel HIP-event means; with ALT="VAR=VALUE" every batch is also rendered with that launch-time switch set, so that
a kernel variant is compared ON THE SAME physical pages.  (GPU box)"""
import os
os.environ.setdefault("WAA_USE_MEASURE_LIB", "1")  # A/B and probe tools flip measurement switches: libwaa_hip_measure.so
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
import web_audio_api_rs_amd as waa  # noqa: E402


def timed(ctx, iters=10):
    ctx.profile_reset()
    for _ in range(iters):
        ctx.render_async()
    ctx.sync()
    return {n: round(ms / iters, 3) for n, l, ms in ctx.profile_entries()}


def main(argv):
    name = argv[0] if argv else "c2"
    n_inst, frames = 1024, 480000
    hip = waa.default_binding()
    noise = torch.empty((n_inst, 2, frames), dtype=torch.float32, device="cuda").uniform_(-1, 1)
    alt = os.environ.get("ALT", "")
    spacers = []
    for trial in range(int(os.environ.get("TRIALS", "14"))):
        if trial % 2 == 1:  # perturb the allocator: a spacer of an odd size stays alive
            spacers.append(torch.empty(((trial * 37 + 11) << 20,), dtype=torch.uint8, device="cuda"))
        ctx, _ = bench.build_workload(waa, hip, name, n_inst, frames, 0, noise.data_ptr())
        ctx.prepare()
        ctx.render_async()
        ctx.sync()
        ctx.profile(True)
        line = [f"{name} trial {trial} out {ctx.output_device()[0]:#x}", str(timed(ctx))]
        if alt:
            var, _, value = alt.partition("=")
            for rep in range(2):
                os.environ[var] = value
                line.append(f"{alt}: {timed(ctx)}")
                os.environ.pop(var)
                line.append(f"plain: {timed(ctx)}")
        print(" ".join(line), flush=True)
        ctx.close()


if __name__ == "__main__":
    main(sys.argv[1:])
