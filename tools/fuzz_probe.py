#!/usr/bin/env python
"""tools/fuzz_probe.py <seed> [--probe] [--frozen] [--live] — render one graph of tests/test_fuzz_graphs.py on the HIP path and on the oracle and
report where they differ; with --probe every node is tapped in turn (its output alone connected to the destination), which
localises a divergence to the first node whose output differs.  Debugging aid for the planner / dyn_kernel (GPU box)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import web_audio_api_rs_amd as waa  # noqa: E402
from test_fuzz_graphs import build_random_graph as _build  # noqa: E402

waa.set_hrtf_database(os.path.join(ROOT, "tests", "golden", "IRC_1003_C.bin"))


def build_random_graph(be, seed):
    return _build(be, seed, frozen="--frozen" in sys.argv)  # (the generator with oversampled shapers / HRTF panners)


def render(be, seed, probe=None, want_plan=False, count_probe=False):
    c, descr = build_random_graph(be, seed)
    if probe is not None:
        feeders = [e[0] for e in c._edges if e[2] == 0]
        c._edges = [e for e in c._edges if e[2] != 0]
        if "--live" in sys.argv:  # keep every node of the full graph live: former destination feeders go through Gain(0)
            for f in feeders:
                c._nodes[f].connect(c.create_gain(gain=0.0)).connect(c.destination())
        if count_probe:  # a StereoPanner behind the node: its mono / stereo law makes the node's channel COUNT audible
            c._nodes[probe].connect(c.create_stereo_panner(pan=0.37)).connect(c.destination())
        else:
            c._nodes[probe].connect(c.destination())
    plan = c.plan_describe() if want_plan else ""
    out = c.start_rendering_sync().data
    c.close()
    return out, descr, plan


def report(tag, g, o):
    d = np.abs(g.astype(np.float64) - o)
    rms = np.sqrt((d ** 2).mean(axis=-1))
    bad = np.argwhere(d > 1e-5 * max(1.0, np.abs(o).max()))
    first = {}
    for i, c, f in bad:
        first.setdefault((int(i), int(c)), int(f))
    print(f"{tag}: max|d| {d.max():.3g} rms {rms.max():.3g} first divergent frame per (inst, ch): "
          f"{ {k: (v, v // 128) for k, v in sorted(first.items())} }")


def main():
    seed = int(sys.argv[1])
    hip = waa.default_binding()
    orc = waa.bind(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")), "orc_")
    g, descr, plan = render(hip, seed, want_plan=True)
    o, _, _ = render(orc, seed)
    print(descr)
    print(plan)
    report("destination", g, o)
    if "--probe" in sys.argv:
        c, _ = build_random_graph(hip, seed)
        n_nodes = len(c._nodes)
        names = [type(n).__name__ for n in c._nodes]
        edges = list(c._edges)
        c.close()
        print(edges)
        for k in range(1, n_nodes):
            for cp in (False, True):
                try:
                    g, _, _ = render(hip, seed, probe=k, count_probe=cp)
                    o, _, _ = render(orc, seed, probe=k, count_probe=cp)
                    report(f"node {k} {names[k]}{' (count probe)' if cp else ''}", g, o)
                except waa.WaaError as e:
                    print(f"node {k} {names[k]}: {e}")


if __name__ == "__main__":
    main()
