"""Same-process A/B of LIBRARY BUILDS: python tools/ab_lib.py alt1.so[,alt2.so...] workload [workload ...]
Renders each bench.py workload alternately with csrc/libwaa_hip.so and with the alternative builds (e.g. waa_conv.hip
compiled with -DWAA_CONV_POL=3), AB_REPS times over so that drift shows, and prints the per-kernel HIP-event means.
For compile-time choices that have no run-time switch (tools/ab_env.py covers those that do).  (GPU box)"""
import ctypes
import os
os.environ.setdefault("WAA_USE_MEASURE_LIB", "1")  # A/B and probe tools flip measurement switches: libwaa_hip_measure.so
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
import web_audio_api_rs_amd as waa  # noqa: E402


def main(argv):
    n_inst, frames = 1024, 480000
    libs = [("default", waa.default_binding())]
    for path in argv[0].split(","):
        libs.append((os.path.basename(path), waa.bind(ctypes.CDLL(os.path.abspath(path)), "waa_")))
    noise = torch.empty((n_inst, 2, frames), dtype=torch.float32, device="cuda").uniform_(-1, 1)
    iters = int(os.environ.get("AB_ITERS", "5"))
    for name in argv[1:]:
        for rep in range(int(os.environ.get("AB_REPS", "2"))):
            for tag, binding in libs:
                ctx, _ = bench.build_workload(waa, binding, name, n_inst, frames, 0, noise.data_ptr())
                ctx.prepare()
                ctx.render_async()
                ctx.sync()
                ctx.profile(True)
                ctx.profile_reset()
                for _ in range(iters):
                    ctx.render_async()
                ctx.sync()
                print(name, tag, {n: round(ms / iters, 3) for n, l, ms in ctx.profile_entries()}, flush=True)
                ctx.close()


if __name__ == "__main__":
    main(sys.argv[1:])
