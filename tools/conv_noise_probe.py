#!/usr/bin/env python
"""tools/conv_noise_probe.py [contexts] [seconds] — what the convolver's noise floor costs (GPU box): a stereo source
that ends half-way -> ConvolverNode (3000 taps) -> DelayNode -> destination, a dynamic plan; per-kernel totals with the floor (default)
and, under WAA_NO_CONV_NOISE_FLOOR=1 (measurement build), with round 3's clearing only.  The slot `conv_code_kernel` holds the three
launches around the code kernel (conv_nz_kernel, conv_code_kernel, conv_floor_kernel)."""
import os
os.environ.setdefault("WAA_USE_MEASURE_LIB", "1")  # A/B and probe tools flip measurement switches: libwaa_hip_measure.so
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import web_audio_api_rs_amd as waa  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
SR = 48000.0
frames = int(secs * SR)
rng = np.random.default_rng(0)
ctx = waa.OfflineAudioContext(2, frames, SR, n_instances=n, binding=waa.default_binding())
src = ctx.create_buffer_source()
src.set_buffer_batch(rng.uniform(-1, 1, (n, 2, frames // 2)).astype(np.float32), SR)
ir = (rng.standard_normal((2, 3000)) * np.exp(-np.arange(3000) / 600.0)).astype(np.float32) * 0.05
conv = ctx.create_convolver(buffer=waa.AudioBuffer(ir, SR), disable_normalization=True)
d = ctx.create_delay(1.0, delay_time=0.02)
src.connect(conv)
conv.connect(d)
d.connect(ctx.destination())
src.start()
for line in ctx.plan_describe().splitlines():
    if "dynamic" in line or "convolver" in line:
        print(line[:200])
ctx.profile(True)
t0 = time.time()
ctx.render_async()
ctx.sync()
print("first render (plan + kernels) ms: %.2f" % ((time.time() - t0) * 1e3))
total = 0.0
for name, launches, ms in ctx.profile_entries():
    if launches:
        total += ms
        print("  %-28s launches %4d  total %.3f ms" % (name, launches, ms))
print("  sum of kernels %.3f ms" % total)
