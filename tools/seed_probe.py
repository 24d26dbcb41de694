#!/usr/bin/env python3
"""tools/seed_probe.py <seed> [1 = the frozen-state generator] — one graph of tests/test_fuzz_graphs.py on the HIP library against the
oracle under the planner's main switches (measurement build): which plan form carries a mismatch (GPU box).  Then
tools/fuzz_probe.py <seed> --probe --live and tools/code_diff.py <seed> localise it to a node and a quantum."""
import os, sys
os.environ["WAA_USE_MEASURE_LIB"] = "1"
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import ctypes, glob
import numpy as np
import web_audio_api_rs_amd as waa
from test_fuzz_graphs import build_random_graph
seed = int(sys.argv[1]); frozen = len(sys.argv) > 2 and sys.argv[2] == "1"
orc = waa.bind(ctypes.CDLL(glob.glob('/root/repo/oracle/*.so')[0]), "orc_")
co, _ = build_random_graph(orc, seed, frozen=frozen)
o = co.start_rendering_sync().data; co.close()
hip = waa.default_binding()
for env in [None, "WAA_NO_SHORT_RING", "WAA_NO_ECHO_RING", "WAA_NO_ECHO_FF", "WAA_NO_ECHO_TAIL", "WAA_NO_DELAY_FOLD", "WAA_NO_LOOP_FOLD", "WAA_NO_FM_FOLD", "WAA_NO_LFO_FOLD", "WAA_NO_ECHO_BQ", "WAA_OSC_EXACT", "WAA_STATIC_CHANNEL_COUNTS"]:
    if env: os.environ[env] = "1"
    try:
        ch, d = build_random_graph(hip, seed, frozen=frozen)
        plan = ch.plan_describe()
        g = ch.start_rendering_sync().data; ch.close()
        print(env, "max", float(np.abs(g - o).max()), "dynamic" if "dynamic-count group" in plan else "static")
        if env is None:
            print(d); print(plan)
    except Exception as e:
        print(env, "ERR", repr(e)[:200])
    if env: del os.environ[env]
