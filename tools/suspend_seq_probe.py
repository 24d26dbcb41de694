"""tools/suspend_seq_probe.py first last — render the suspend-fuzz graphs of seeds [first, last] one after the other in ONE process (stale
device memory from the earlier ones) and compare the last with the oracle: where does it differ?  (GPU box)"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import web_audio_api_rs_amd as waa
waa.set_hrtf_database(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "IRC_1003_C.bin"))
from test_fuzz_graphs import build_random_graph
from test_fuzz_suspend import mutate
hip = waa.default_binding()
orc = waa.bind(ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "liboracle.so")), "orc_")
first, last = int(sys.argv[1]), int(sys.argv[2])
for seed in range(first, last + 1):
    ch, descr = build_random_graph(hip, seed)
    try:
        edits = mutate(ch, seed)
        plan = ch.plan_describe()
        g = ch.start_rendering_sync().data
    except waa.WaaError as e:
        ch.close()
        continue
    ch.close()
co, _ = build_random_graph(orc, last)
mutate(co, last)
o = co.start_rendering_sync().data
d = np.abs(g.astype(np.float64) - o)
print(last, descr, "|", edits, "max", d.max())
for i in range(d.shape[0]):
    for c in range(d.shape[1]):
        bad = np.nonzero(~(d[i, c] < 1e-4))[0]
        if bad.size:
            print(f"  inst {i} ch {c}: {bad.size} frames differ, first {bad[0]} (quantum {bad[0] // 128}, tile {bad[0] // 2048}), last {bad[-1]}; device there {g[i, c, bad[:4]]}")
print(plan)
