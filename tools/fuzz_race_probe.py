#!/usr/bin/env python
"""tools/fuzz_race_probe.py FIRST COUNT JOBS — debugging aid (GPU box): J processes render the same seed range of the random-graph
generator side by side and print WHERE a graph differs from the oracle (instance, channel, first / last frame) — for failures that
only show up when several processes share the device."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(first, count):
    import numpy as np
    import web_audio_api_rs_amd as waa
    from test_fuzz_graphs import build_random_graph
    hip = waa.default_binding()
    orc = waa.bind(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")), "orc_")
    for seed in range(first, first + count):
        try:
            ch, descr = build_random_graph(hip, seed)
            if "auto-biquad" not in str(descr):
                ch.close()
                continue
            plan = ch.plan_describe()
            g = ch.start_rendering_sync().data
            ch.close()
        except waa.WaaError:
            continue
        co, _ = build_random_graph(orc, seed)
        o = co.start_rendering_sync().data
        co.close()
        d = np.abs(g - o)
        if d.max() > 2e-5 * max(1.0, np.abs(o).max()):
            print("MISMATCH seed", seed, descr, "max", float(d.max()), flush=True)
            for i in range(g.shape[0]):
                for c in range(g.shape[1]):
                    bad = np.nonzero(d[i, c] > 1e-5)[0]
                    if len(bad):
                        print("  inst", i, "ch", c, "n_bad", len(bad), "first", int(bad[0]), "last", int(bad[-1]), "g/o", float(g[i, c, bad[0]]),
                              float(o[i, c, bad[0]]), flush=True)
            print(plan, flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), int(sys.argv[3]))
    else:
        first, count, jobs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(first), str(count)]) for _ in range(jobs)]
        for p in ps:
            p.wait()
