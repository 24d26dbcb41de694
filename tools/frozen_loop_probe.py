#!/usr/bin/env python
"""tools/frozen_loop_probe.py [contexts] [seconds] — time a feedback echo with an oversampled WaveShaper (and one with an HRTF panner)
in the loop: rendered quantum block by quantum block (ranged launches) — a correctness path, this says what it costs."""
import os
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
waa.set_hrtf_database(os.path.join(ROOT, "tests", "golden", "IRC_1003_C.bin"))
rng = np.random.default_rng(0)
noise = rng.uniform(-1, 1, (n, 2, frames)).astype(np.float32)
for kind in ("2x", "hrtf"):
    ctx = waa.OfflineAudioContext(2, frames, SR, n_instances=n, binding=waa.default_binding())
    src = ctx.create_buffer_source()
    src.set_buffer_batch(noise, SR)
    d = ctx.create_delay(1.0, delay_time=0.25)
    f = ctx.create_wave_shaper(curve=np.tanh(np.linspace(-2, 2, 129)).astype(np.float32), oversample="2x") if kind == "2x" else \
        ctx.create_panner(panning_model="HRTF", position=(1.0, 0.3, -0.5))
    src.connect(d)
    d.connect(f).connect(ctx.create_gain(gain=0.4)).connect(d)
    f.connect(ctx.destination())
    src.start()
    print([l for l in ctx.plan_describe().splitlines() if "loop" in l][-1])
    ctx.profile(True)
    t0 = time.time()
    ctx.render_async()
    ctx.sync()
    print("%s in the loop, %d contexts x %g s: first render %.1f ms" % (kind, n, secs, (time.time() - t0) * 1e3))
    t0 = time.time()
    ctx.render_async()
    ctx.sync()
    print("   second render %.1f ms" % ((time.time() - t0) * 1e3))
    for name, launches, ms in ctx.profile_entries():
        if launches:
            print("   %-24s launches %6d  total %8.3f ms" % (name, launches, ms))
    ctx.close()
