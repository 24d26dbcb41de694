"""tools/c2_variants_probe.py — C2's streaming Biquad, every kernel form on the SAME batch (launch-time switches of the measurement
build), for several fresh batches, on plain hipMalloc and out of a graded arena: is a slow batch slow in its copy-only form too
(memory), or only with the arithmetic (issue / overlap)?  (GPU box)"""
import os
os.environ.setdefault("WAA_USE_MEASURE_LIB", "1")
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
import web_audio_api_rs_amd as waa  # noqa: E402

VARIANTS = [("product", None), ("copy-only", ("WAA_STREAM_DEBUG", "1")), ("no-stores", ("WAA_STREAM_DEBUG", "2")),
            ("digest", ("WAA_BIQUAD_DIGEST", "1")), ("prefetch1", ("WAA_STREAM_PREFETCH1", "1")), ("product", None)]
EXTRA = [v for v in os.environ.get("C2_EXTRA", "").split(",") if v]


def timed(ctx, iters=8):
    ctx.profile(True)
    ctx.profile_reset()
    for _ in range(iters):
        ctx.render_async()
    ctx.sync()
    return sum(ms / max(l, 1) for n, l, ms in ctx.profile_entries())


def batches(hip, noise, tag, n):
    for k in range(n):
        ctx, _ = bench.build_workload(waa, hip, "c2", 1024, 480000, 0, noise.data_ptr())
        ctx.prepare()
        ctx.render_async()
        ctx.sync()
        cells = []
        for label, sw in VARIANTS + [(e, (e, "1")) for e in EXTRA]:
            if sw:
                os.environ[sw[0]] = sw[1]
            cells.append(f"{label} {timed(ctx):.3f}")
            if sw:
                os.environ.pop(sw[0])
        print(f"{tag} batch {k} out {ctx.output_device()[0]:#x}: " + " | ".join(cells), flush=True)
        ctx.close()


def main():
    hip = waa.default_binding()
    noise = torch.empty((1024, 2, 480000), dtype=torch.float32, device="cuda").uniform_(-1, 1)
    batches(hip, noise, "plain", int(os.environ.get("N_PLAIN", "4")))
    hip.check(hip.device_arena_reserve_graded(0, int(os.environ.get("ARENA_GB", "32")) << 30, int(os.environ.get("CAND_GB", "100")) << 30))
    print({k: v for k, v in waa.arena_grades(hip, 0).items() if k != "unit_ms"}, flush=True)
    batches(hip, noise, "arena", int(os.environ.get("N_ARENA", "6")))
    hip.check(hip.device_arena_reserve(0, 0))


if __name__ == "__main__":
    main()
