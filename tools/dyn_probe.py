#!/usr/bin/env python
"""tools/dyn_probe.py [contexts] [seconds] — time a graph the planner renders with exact per-quantum channel counts
(dyn_kernel): a mono source from 0 s plus a stereo source from 1 s into Biquad -> StereoPanner (GPU box)."""
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
a = ctx.create_buffer_source()
a.set_buffer_batch(rng.uniform(-1, 1, (n, 1, frames)).astype(np.float32), SR)
b = ctx.create_buffer_source()
b.set_buffer_batch(rng.uniform(-1, 1, (n, 2, frames // 2)).astype(np.float32), SR)
bq = ctx.create_biquad_filter(type_="lowpass", frequency=2000.0)
pan = ctx.create_stereo_panner(pan=0.3)
a.connect(bq)
b.connect(bq)
bq.connect(pan).connect(ctx.destination())
a.start()
b.start_at(1.0)
print(ctx.plan_describe())
ctx.profile(True)
t0 = time.time()
ctx.render_async()
ctx.sync()
t1 = time.time()
print("first render (plan + kernels) ms: %.2f" % ((t1 - t0) * 1e3))
for name, launches, ms in ctx.profile_entries():
    if launches:
        print("  %-28s launches %4d  total %.3f ms" % (name, launches, ms))
