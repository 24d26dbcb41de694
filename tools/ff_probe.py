"""debugging aid: where the feed-forward ring echo differs from the oracle (python tools/ff_probe.py)"""
import os
os.environ.setdefault("WAA_USE_MEASURE_LIB", "1")  # A/B and probe tools flip measurement switches: libwaa_hip_measure.so, sys, ctypes
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["WAA_ECHO_FF_MIN_INST"] = "1"
import numpy as np
import web_audio_api_rs_amd as waa
import test_delay as td
from graphs import white_noise
hip = waa.default_binding()
orc = waa.bind(ctypes.CDLL([os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle")) if f.endswith(".so")][0]), "orc_")
n, length = 6, 2048 * 11 + 77
noise = white_noise(n, 2, length, seed0=41)
for variant in ("dry+wet", "wet"):
    g, plan = td._ff_echo(hip, noise, td.FF_DELAYS, variant, length=length)
    o, _ = td._ff_echo(orc, noise, td.FF_DELAYS, variant, length=length)
    d = np.abs(g - o)
    print(variant, "max", d.max(), "ring" if "nothing fed back" in plan else "plain", "\n".join(l[:200] for l in plan.splitlines()[1:]))
    for i in range(n):
        for c in range(2):
            bad = np.nonzero(d[i, c])[0]
            if len(bad):
                print("  inst", i, "ch", c, "n_bad", len(bad), "first", bad[:6], "last", bad[-3:], "g", g[i, c, bad[0]], "o", o[i, c, bad[0]],
                      "dry", noise[i, c, bad[0]])
