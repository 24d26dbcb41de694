#!/usr/bin/env python
"""tools/fuzz_repeat_probe.py FROZEN SEED REPEATS JOBS — debugging aid (GPU box): J processes render ONE random graph REPEATS times
each, side by side, and report every render that differs from the oracle's (rare, load-dependent failures)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(frozen, seed, repeats):
    import numpy as np
    import web_audio_api_rs_amd as waa
    from test_fuzz_graphs import build_random_graph
    waa.set_hrtf_database(os.path.join(ROOT, "tests", "golden", "IRC_1003_C.bin"))
    hip = waa.default_binding()
    orc = waa.bind(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")), "orc_")
    co, descr = build_random_graph(orc, seed, frozen=frozen)
    o = co.start_rendering_sync().data
    co.close()
    bad_runs = 0
    for r in range(repeats):
        ch, _ = build_random_graph(hip, seed, frozen=frozen)
        g = ch.start_rendering_sync().data
        ch.close()
        d = np.abs(g - o)
        if d.max() > 2e-5 * max(1.0, np.abs(o).max()):
            bad_runs += 1
            if bad_runs <= 2:
                print("MISMATCH run", r, "seed", seed, descr, "max", float(d.max()), flush=True)
                for i in range(g.shape[0]):
                    for c in range(g.shape[1]):
                        bad = np.nonzero(d[i, c] > 1e-5)[0]
                        if len(bad):
                            print("  inst", i, "ch", c, "n_bad", len(bad), "first", int(bad[0]), "q", int(bad[0]) // 128, "last", int(bad[-1]),
                                  float(g[i, c, bad[0]]), float(o[i, c, bad[0]]), flush=True)
    print("worker done: bad runs", bad_runs, "of", repeats, flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--worker":
        worker(sys.argv[2] == "1", int(sys.argv[3]), int(sys.argv[4]))
    else:
        frozen, seed, repeats, jobs = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", frozen, seed, repeats]) for _ in range(jobs)]
        for p in ps:
            p.wait()
