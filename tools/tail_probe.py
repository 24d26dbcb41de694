#!/usr/bin/env python
"""tools/tail_probe.py [--frozen] SEED INST — debugging aid (GPU box): for every node of the random graph SEED (tests/test_fuzz_graphs.py,
rendered once per node with only that node connected to the destination, like fuzz_tap_probe.py) the LAST frame of instance INST that
holds a non-zero / a normal f32 sample, on the device and on the oracle — where does a tail end one quantum apart?"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import web_audio_api_rs_amd as waa  # noqa: E402
from test_fuzz_graphs import build_random_graph  # noqa: E402

frozen = "--frozen" in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith("--")]
seed, inst = int(args[0]), int(args[1])
waa.set_hrtf_database(os.path.join(ROOT, "tests", "golden", "IRC_1003_C.bin"))
hip = waa.default_binding()
orc = waa.bind(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")), "orc_")
TINY = np.finfo(np.float32).tiny


def last(x, pred):
    idx = np.nonzero(pred(x))[0]
    return int(idx[-1]) if idx.size else -1


for tap in range(64):
    ch, descr = build_random_graph(hip, seed, frozen=frozen, tap=tap)
    if ch is None:
        break
    try:
        g = ch.start_rendering_sync().data[inst]
    except waa.WaaError as e:
        print("tap", tap, "refused:", str(e)[:80])
        ch.close()
        continue
    ch.close()
    co, _ = build_random_graph(orc, seed, frozen=frozen, tap=tap)
    o = co.start_rendering_sync().data[inst]
    co.close()
    for name, x in (("device", g), ("oracle", o)):
        nz = [last(x[c], lambda v: v != 0) for c in range(x.shape[0])]
        nm = [last(x[c], lambda v: np.abs(v) >= TINY) for c in range(x.shape[0])]
        k = max(nz)
        tail = x[:, max(0, k - 3):k + 2] if k >= 0 else x[:, :0]
        print(f"tap {tap} {descr.split('|')[-1].strip()[:40]:40s} {name}: last non-zero frame {nz} (quantum {[n // 128 for n in nz]}), last normal {nm}, values there {np.array2string(tail, precision=3)}")
