import ctypes, os, sys
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tests")
import numpy as np
import web_audio_api_rs_amd as waa
from test_fuzz_graphs import build_random_graph
waa.set_hrtf_database(ROOT+"/tests/golden/IRC_1003_C.bin")
hip = waa.default_binding(); orc = waa.bind(ctypes.CDLL(ROOT+"/oracle/liboracle.so"), "orc_")
seed, tap = int(sys.argv[1]), int(sys.argv[2])
ch, d = build_random_graph(hip, seed, frozen=True, tap=tap)
print(ch.plan_describe())
g = ch.start_rendering_sync().data; ch.close()
co,_ = build_random_graph(orc, seed, frozen=True, tap=tap); o = co.start_rendering_sync().data; co.close()
dd=np.abs(g-o); print("max", dd.max())
i=0
for f in (250,255,256,257,300,600,975,980):
    print(f, g[i,:,f], o[i,:,f])
