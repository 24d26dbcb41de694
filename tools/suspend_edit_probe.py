"""tools/suspend_edit_probe.py <seed> — a suspend-fuzz graph (tests/test_fuzz_suspend.py) with each of its edits left out in turn: which
edit does a mismatch hinge on?  (GPU box)"""
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
seed = int(sys.argv[1])


def render(be, skip):
    ch, descr = build_random_graph(be, seed)
    real = ch.suspend_sync
    count = [0]

    def filtered(t, cb):
        k = count[0]
        count[0] += 1
        if k in skip:
            return
        return real(t, cb)
    ch.suspend_sync = filtered
    edits = mutate(ch, seed)
    nodes = [(n.id, type(n).__name__) for n in ch._nodes]
    out = ch.start_rendering_sync().data
    ch.close()
    return out, edits, descr, nodes


_, edits, descr, nodes = render(orc, set())
print(descr, "|", edits)
print(nodes)
n_edits = len(edits.split("+"))
for skip in [set()] + [{k} for k in range(n_edits)]:
    g, _, _, _ = render(hip, skip)
    o, _, _, _ = render(orc, skip)
    d = np.abs(g.astype(np.float64) - o)
    first = {(i, c): int(np.argmax(d[i, c] > 1e-5)) // 128 for i in range(d.shape[0]) for c in range(d.shape[1]) if (d[i, c] > 1e-5).any()}
    print("without edit(s)", sorted(skip), "max", float(d.max()), "first divergent quantum", first)
