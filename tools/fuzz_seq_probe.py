import ctypes, os, sys
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
import numpy as np
import web_audio_api_rs_amd as waa
from test_fuzz_graphs import build_random_graph
waa.set_hrtf_database(os.path.join(ROOT, "tests", "golden", "IRC_1003_C.bin"))
hip = waa.default_binding(); orc = waa.bind(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")), "orc_")
frozen = sys.argv[1] == "1"; target = int(sys.argv[2]); back = int(sys.argv[3])
for seed in range(target - back, target + 1):
    try:
        ch, descr = build_random_graph(hip, seed, frozen=frozen)
        g = ch.start_rendering_sync().data; ch.close()
    except waa.WaaError as e:
        continue
    if seed != target: continue
    co,_ = build_random_graph(orc, seed, frozen=frozen); o = co.start_rendering_sync().data; co.close()
    d = np.abs(g-o); print("seed", seed, "back", back, descr, "max", float(d.max()))
    for i in range(g.shape[0]):
        for c in range(g.shape[1]):
            bad = np.nonzero(d[i,c] > 1e-5)[0]
            if len(bad): print("  inst", i, "ch", c, "n_bad", len(bad), "first", int(bad[0]), "q", int(bad[0])//128, "last", int(bad[-1]), float(g[i,c,bad[0]]), float(o[i,c,bad[0]]))
