#!/usr/bin/env python
"""tools/code_diff.py <seed> [--frozen] — per-quantum codes (channel count | 0x80 if silent) of every node on both sides
(WAA_DUMP_CODES / ORC_DUMP_CODES) for one graph of tests/test_fuzz_graphs.py: prints the first quanta where they differ."""
import ctypes
import os
os.environ.setdefault("WAA_USE_MEASURE_LIB", "1")  # A/B and probe tools flip measurement switches: libwaa_hip_measure.so
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import web_audio_api_rs_amd as waa  # noqa: E402
from test_fuzz_graphs import build_random_graph  # noqa: E402

waa.set_hrtf_database(os.path.join(ROOT, "tests", "golden", "IRC_1003_C.bin"))


def load(path):
    raw = np.fromfile(path, dtype=np.uint8)
    n_inst, n_nodes, nq = np.frombuffer(raw[:12].tobytes(), dtype=np.uint32)
    return raw[12:].reshape(n_inst, n_nodes, nq)


def main():
    seed = int(sys.argv[1])
    frozen = "--frozen" in sys.argv
    os.environ["WAA_DUMP_CODES"] = "/tmp/waa_codes.bin"
    os.environ["ORC_DUMP_CODES"] = "/tmp/orc_codes.bin"
    hip = waa.default_binding()
    orc = waa.bind(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")), "orc_")
    outs = []
    for be in (hip, orc):
        c, descr = build_random_graph(be, seed, frozen=frozen)
        names = [type(n).__name__ for n in c._nodes]
        outs.append(c.start_rendering_sync().data)
        c.sync() if be is hip else None
        c.close()
    print(descr)
    g, o = load("/tmp/waa_codes.bin"), load("/tmp/orc_codes.bin")
    for node in range(g.shape[1]):
        for inst in range(g.shape[0]):
            a, b = g[inst, node], o[inst, node]
            if np.all(a == 0xFF):
                continue
            bad = np.nonzero(a != b)[0]
            if bad.size:
                q = int(bad[0])
                print(f"node {node} {names[node]} inst {inst}: first code difference at quantum {q}: device {a[max(0, q - 2):q + 6]} "
                      f"oracle {b[max(0, q - 2):q + 6]} ({bad.size} quanta differ)")
    d = np.abs(outs[0].astype(np.float64) - outs[1])
    print("output max|d| per (inst, ch):", d.max(axis=-1))


if __name__ == "__main__":
    main()
