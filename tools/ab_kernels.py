"""Same-box A/B of library builds: python tools/ab_kernels.py [--workloads c2,c5,echo] libA.so libB.so ...

Every gpurun call lands on a different MI355X box and box-to-box kernel times differ by several percent, so two
builds are only comparable when they run back to back on ONE box.  Build the variants, copy each libwaa_hip.so to
tools/ab/<name>.so (git-ignored, travels with the gpurun snapshot) and pass the paths; the script renders each
bench.py workload five times per library and prints the per-kernel mean (HIP events, ms per render).
A workload name with the suffix -loop forces the quantum-serial loop kernel (WAA_LOOP_KERNEL=1)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
import web_audio_api_rs_amd as waa  # noqa: E402


def main(argv):
    workloads = ["c2", "c5", "c4", "echo"]
    if argv and argv[0] == "--workloads":
        workloads = argv[1].split(",")
        argv = argv[2:]
    n_inst, frames = 1024, 480000
    noise = torch.empty((n_inst, 2, frames), dtype=torch.float32, device="cuda").uniform_(-1, 1)
    for path in argv:
        hip = waa.bind(ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL), "waa_")
        for name in workloads:
            os.environ.pop("WAA_LOOP_KERNEL", None)
            if name.endswith("-loop"):
                os.environ["WAA_LOOP_KERNEL"] = "1"
            ctx, _ = bench.build_workload(waa, hip, name.split("-")[0], n_inst, frames, 0, noise.data_ptr())
            ctx.prepare()
            ctx.render_async()
            ctx.sync()
            ctx.profile(True)
            ctx.profile_reset()
            for _ in range(5):
                ctx.render_async()
            ctx.sync()
            print(os.path.basename(path), name, {n: round(ms / 5, 3) for n, l, ms in ctx.profile_entries()}, flush=True)
            ctx.close()


if __name__ == "__main__":
    main(sys.argv[1:])
