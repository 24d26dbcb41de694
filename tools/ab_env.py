"""Same-process A/B of a library switch: python tools/ab_env.py VAR[=VALUE] workload [workload ...]
Renders each bench.py workload with the environment variable unset and then set (plan-time switches are read when
the batch is planned, launch-time ones at every launch), alternating twice (A B A B) so that drift shows, and prints
the per-kernel HIP-event means.  Example — the linear-prefix fetch of the streaming kernels:
    python tools/ab_env.py WAA_NO_LINEAR_PREFIX c2 iir2 t1"""
import os
os.environ.setdefault("WAA_USE_MEASURE_LIB", "1")  # A/B and probe tools flip measurement switches: libwaa_hip_measure.so
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
import web_audio_api_rs_amd as waa  # noqa: E402


def main(argv):
    var, _, value = argv[0].partition("=")
    value = value or "1"
    n_inst, frames = 1024, 480000
    hip = waa.default_binding()
    noise = torch.empty((n_inst, 2, frames), dtype=torch.float32, device="cuda").uniform_(-1, 1)
    if os.environ.get("ARENA_GB"):  # the batches' buffers out of a graded arena (waa_device_arena_reserve_graded): placement out of the picture
        hip.check(hip.device_arena_reserve_graded(0, int(os.environ["ARENA_GB"]) << 30, int(os.environ.get("CAND_GB", "100")) << 30))
        print({k: v for k, v in waa.arena_grades(hip, 0).items() if k != "unit_ms"}, flush=True)
    for name in argv[1:]:
        for rep in range(int(os.environ.get("AB_REPS", "2"))):
            for on in (False, True):
                os.environ.pop(var, None)
                if on:
                    os.environ[var] = value
                ctx, _ = bench.build_workload(waa, hip, name, n_inst, frames, 0, noise.data_ptr())
                ctx.prepare()
                ctx.render_async()
                ctx.sync()
                ctx.profile(True)
                ctx.profile_reset()
                iters = int(os.environ.get("AB_ITERS", "5"))
                for _ in range(iters):
                    ctx.render_async()
                ctx.sync()
                print(name, f"{var}={'%s' % value if on else '(unset)'}", {n: round(ms / iters, 3) for n, l, ms in ctx.profile_entries()},
                      flush=True)
                ctx.close()
    os.environ.pop(var, None)


if __name__ == "__main__":
    main(sys.argv[1:])
