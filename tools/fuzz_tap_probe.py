#!/usr/bin/env python
"""tools/fuzz_tap_probe.py [--frozen] SEED — debugging aid (GPU box): renders the random graph SEED once per node with only that
node connected to the destination, on the device and on the oracle, and prints where they differ — the first node of the
graph's node list whose output differs is where a mismatch starts."""
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
seed = int([a for a in sys.argv[1:] if not a.startswith("--")][0])
waa.set_hrtf_database(os.path.join(ROOT, "tests", "golden", "IRC_1003_C.bin"))
hip = waa.default_binding()
orc = waa.bind(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")), "orc_")
for tap in range(64):
    ch, descr = build_random_graph(hip, seed, frozen=frozen, tap=tap)
    if ch is None:
        break
    try:
        g = ch.start_rendering_sync().data
    except waa.WaaError as e:
        print("tap", tap, descr.split("|")[-1], "refused:", str(e)[:80])
        ch.close()
        continue
    ch.close()
    co, _ = build_random_graph(orc, seed, frozen=frozen, tap=tap)
    o = co.start_rendering_sync().data
    co.close()
    d = np.abs(g - o)
    line = "tap %d %s max %.3g" % (tap, descr.split("|")[-1], float(d.max()))
    for i in range(g.shape[0]):
        for c in range(g.shape[1]):
            bad = np.nonzero(d[i, c] > 1e-5)[0]
            if len(bad):
                line += " [inst %d ch %d n %d first %d last %d]" % (i, c, len(bad), bad[0], bad[-1])
    print(line)
