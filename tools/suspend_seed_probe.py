import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import web_audio_api_rs_amd as waa
import ctypes
waa.set_hrtf_database("/root/repo/tests/golden/IRC_1003_C.bin")
from test_fuzz_graphs import build_random_graph, FRAMES
from test_fuzz_suspend import mutate
hip = waa.default_binding()
orc = waa.bind(ctypes.CDLL("/root/repo/oracle/liboracle.so"), "orc_")
for seed in map(int, sys.argv[1:]):
    ch, descr = build_random_graph(hip, seed)
    edits = mutate(ch, seed)
    plan = ch.plan_describe()
    g = ch.start_rendering_sync().data
    co, _ = build_random_graph(orc, seed)
    mutate(co, seed)
    o = co.start_rendering_sync().data
    d = np.abs(g.astype(np.float64) - o)
    first = {(i, c): int(np.argmax(d[i, c] > 1e-5)) // 128 for i in range(d.shape[0]) for c in range(d.shape[1]) if (d[i, c] > 1e-5).any()}
    print(seed, descr, "|", edits, "max", d.max(), "first divergent quantum", first)
    print("\n".join(l for l in plan.splitlines() if "suspend" in l or "loop" in l or "dynamic" in l)[:1500])
