#!/usr/bin/env python
"""tools/last_normal_probe.py <seed> [--frozen] <node> ... — for every given node of a fuzz graph, tapped alone: the last render quantum
per instance that holds a NORMAL f32 value (what DelayReader tests, delay.rs:640-660), on the HIP path and on the oracle, and the values
around that quantum.  Debugging aid for data-dependent silence (GPU box)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

import fuzz_probe as fp  # noqa: E402
import web_audio_api_rs_amd as waa  # noqa: E402


def main():
    seed = int(sys.argv[1])
    nodes = [int(a) for a in sys.argv[2:] if a.isdigit()]
    hip = waa.default_binding()
    orc = waa.bind(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")), "orc_")
    tiny = np.finfo(np.float32).tiny
    for k in nodes:
        g, _, _ = fp.render(hip, seed, probe=k)
        o, _, _ = fp.render(orc, seed, probe=k)
        for name, a in (("hip", g), ("orc", o)):
            a = np.asarray(a, dtype=np.float32)
            nq = a.shape[2] // 128
            act = (np.abs(a[:, :, :nq * 128]) >= tiny).reshape(a.shape[0], a.shape[1], nq, 128).any(axis=(1, 3))
            last = [int(np.nonzero(r)[0].max()) if r.any() else -1 for r in act]
            sub = ((np.abs(a) > 0) & (np.abs(a) < tiny)).sum()
            print(f"node {k} {name}: last quantum with a normal value per instance {last}; subnormal values in the render: {int(sub)}")
        g = np.asarray(g, dtype=np.float32)
        o = np.asarray(o, dtype=np.float32)
        for i in range(g.shape[0]):
            d = np.nonzero((g[i] != o[i]).any(axis=0))[0]
            if len(d):
                f = int(d[0])
                print(f"  inst {i}: first unequal frame {f} (quantum {f // 128}): hip {g[i, :, f]} orc {o[i, :, f]}; "
                      f"max|hip| there-after {np.abs(g[i, :, f:]).max():.3g} max|orc| {np.abs(o[i, :, f:]).max():.3g}")


if __name__ == "__main__":
    main()
