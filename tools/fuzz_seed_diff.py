"""Where does a random graph of tests/test_fuzz_graphs.py differ from the oracle?  python tools/fuzz_seed_diff.py SEED...
(GPU box).  Prints, per instance and channel, how many frames are off by more than 1e-5, the first and last of them
and the two values there — enough to tell a start / end / quantum-shift problem from a numerical one.  The oracle is
test infrastructure; this tool is a debugging aid for the tests, not part of the product."""
import ctypes
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import web_audio_api_rs_amd as waa
from test_fuzz_graphs import build_random_graph
waa.set_hrtf_database(os.path.join(ROOT, "tests", "golden", "IRC_1003_C.bin"))
hip = waa.default_binding(); orc = waa.bind(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")), "orc_")
FROZEN = "--frozen" in sys.argv   # the generator with oversampled WaveShapers / HRTF panners
for seed in map(int, [a for a in sys.argv[1:] if not a.startswith("--")]):
    ch, descr = build_random_graph(hip, seed, frozen=FROZEN)
    if "--plan" in sys.argv:
        print(ch.plan_describe())
    g = ch.start_rendering_sync().data
    co, _ = build_random_graph(orc, seed, frozen=FROZEN)
    o = co.start_rendering_sync().data
    d = np.abs(g - o)
    print("seed", seed, descr, "max", float(d.max()))
    for i in range(g.shape[0]):
        for c in range(g.shape[1]):
            bad = np.nonzero(d[i, c] > 1e-5)[0]
            if len(bad):
                print("  inst", i, "ch", c, "n_bad", len(bad), "first", int(bad[0]), "q", int(bad[0]) // 128, "last", int(bad[-1]), "g/o at first", float(g[i,c,bad[0]]), float(o[i,c,bad[0]]))
