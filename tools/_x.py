import sys, ctypes
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import web_audio_api_rs_amd as waa
import bench
n_inst, frames, SR = 1024, 480000, 48000.0
noise = torch.empty((n_inst, 2, frames), dtype=torch.float32, device="cuda").uniform_(-1, 1)
for path in sys.argv[1:]:
    hip = waa.bind(ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL), "waa_")
    for name in ("c2", "c2k", "iir2", "iir8", "c5", "echo", "fb", "fb-loop"):
        import os
        os.environ.pop('WAA_LOOP_KERNEL', None)
        if name.endswith('-loop'): os.environ['WAA_LOOP_KERNEL'] = '1'
        ctx, _ = bench.build_workload(waa, hip, name.split('-')[0], n_inst, frames, 0, noise.data_ptr())
        ctx.prepare(); ctx.render_async(); ctx.sync()
        ctx.profile(True); ctx.profile_reset()
        for _ in range(5): ctx.render_async()
        ctx.sync()
        print(path[-7:], name, {n: round(ms/max(l,1)*(l/5),3) for n,l,ms in ctx.profile_entries()}, flush=True)
        ctx.close()
