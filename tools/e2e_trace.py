#!/usr/bin/env python3
"""tools/e2e_trace.py [--pcm] — one waa_render_sharded call of the C2 graph from pinned host buffers with WAA_SHARD_TRACE=1:
the phase timeline of every sub-batch on stderr (GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["WAA_SHARD_TRACE"] = "1"
import torch  # noqa: E402
import bench  # noqa: E402
import web_audio_api_rs_amd as waa  # noqa: E402
from web_audio_api_rs_amd.sharding import render_sharded  # noqa: E402

pcm = "--pcm" in sys.argv
arena = "--arena" in sys.argv
sub = int(sys.argv[sys.argv.index("--sub") + 1]) if "--sub" in sys.argv else 8
if "--quiet" in sys.argv:
    os.environ.pop("WAA_SHARD_TRACE", None)
n, frames = 1024, 480000
hip = waa.default_binding()
if pcm:
    host_in = torch.empty((n, frames, 2), dtype=torch.int16, pin_memory=True).random_(-32768, 32767)
    host_out = torch.empty((n, frames, 2), dtype=torch.int16, pin_memory=True)
else:
    host_in = torch.empty((n, 2, frames), dtype=torch.float32, pin_memory=True).uniform_(-1.0, 1.0)
    host_out = torch.empty((n, 2, frames), dtype=torch.float32, pin_memory=True)


def build(n_sub_inst, device):
    return bench.build_workload(waa, hip, "c2", n_sub_inst, frames, device, None)


if arena:
    hip.check(hip.device_arena_reserve(0, 24 << 30))
for rep in range(4):
    sys.stderr.write(f"---- pass {rep}\n")
    r = render_sharded(build, host_in, host_out, devices=(0,), sub_batches=sub, sample_rate=48000.0, pcm16=pcm, out_pcm16=pcm)
    sys.stderr.write(f"---- pass {rep}: {r['seconds'] * 1e3:.1f} ms\n")
