"""tools/overlap_probe.py — do the three convolver kernels of T1 overlap usefully when two half-size batches render on two streams?
(forward transform + Biquad: issue-bound, product: memory-bound, inverse: issue-bound.)  Prints sequential vs concurrent time of
two 512-context T1 batches and the time of one 1024-context batch.  (GPU box)"""
import os
os.environ.setdefault("WAA_USE_MEASURE_LIB", "1")  # A/B and probe tools flip measurement switches: libwaa_hip_measure.so
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
import web_audio_api_rs_amd as waa  # noqa: E402

frames = 480000
hip = waa.default_binding()
name = sys.argv[1] if len(sys.argv) > 1 else "t1"


def make(n):
    noise = torch.empty((n, 2, frames), dtype=torch.float32, device="cuda").uniform_(-1, 1)
    ctx, _ = bench.build_workload(waa, hip, name, n, frames, 0, noise.data_ptr())
    ctx.prepare()
    ctx.render_async()
    ctx.sync()
    return ctx, noise


def wall(fn, reps=8):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best


full, _n0 = make(1024)
a, _n1 = make(512)
b, _n2 = make(512)
c, _n3 = make(256)
d, _n4 = make(256)


def seq():
    a.render_async(); a.sync(); b.render_async(); b.sync()


def conc():
    a.render_async(); b.render_async(); a.sync(); b.sync()


def conc4():
    for x in (a, b, c, d):
        x.render_async()
    for x in (a, b, c, d):
        x.sync()


def one():
    full.render_async(); full.sync()


print(name, "one batch of 1024: %.3f ms" % wall(one))
print(name, "two batches of 512, one after the other: %.3f ms" % wall(seq))
print(name, "two batches of 512, both in flight: %.3f ms" % wall(conc))
print(name, "512 + 512 + 256 + 256 in flight (1536 contexts): %.3f ms -> per 1024: %.3f" % (wall(conc4), wall(conc4) / 1.5))
