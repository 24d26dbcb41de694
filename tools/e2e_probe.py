#!/usr/bin/env python
"""tools/e2e_probe.py [contexts] — the e2e record of bench.py alone (host buffers in and out, one batch vs sub-batches on host
threads), for A/B of the transfer path (GPU box)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import web_audio_api_rs_amd as waa  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for n_sub in (2, 4, 8):
    rec = bench.e2e_record(torch, waa, waa.default_binding(), n, 10.0, 0, n_sub=n_sub)
    print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in rec.items() if k != "note"}))
