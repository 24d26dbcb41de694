"""N > 1 path on CPU: two gloo ranks shard the contexts (no data-path collective) and the union of their
renders equals the single-process render; bench.py's timing protocol (barrier + MAX over ranks) runs."""
import json
import os
import subprocess
import sys

import numpy as np

import web_audio_api_rs_amd as waa
from graphs import c2, white_noise
from web_audio_api_rs_amd.sharding import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_everything():
    for n in (1, 7, 8, 1024, 4096):
        for world in (1, 2, 3, 8):
            r = [shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def test_two_rank_gloo_shards_match_single_process(orc):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "tests", "_rank_worker.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    got = json.loads(line)
    assert got["ranges"] == [[0, 4], [4, 7]] and got["elapsed"] > 0
    noise = white_noise(7, 2, 128 * 20 + 3)
    ctx, _ = c2(orc, noise)
    ref = ctx.start_rendering_sync().data.astype(np.float64).sum(axis=(1, 2))
    ctx.close()
    assert np.allclose(got["sums"], ref, rtol=0, atol=1e-9)
    assert np.allclose(got["sums_sharded"], ref, rtol=0, atol=1e-9)  # sharding.render_sharded on every rank's shard
