#!/usr/bin/env python
"""tools/pcie_probe.py — host<->device copy bandwidth, one direction at a time and both at once on two streams (pinned host
memory): is the boundary's transfer time (bench.py `e2e`) a sum or a max of upload and download? (GPU box)"""
import time

import torch

n = 1 << 30  # 1 GiB per buffer
h_in = torch.empty(n, dtype=torch.uint8, pin_memory=True)
h_out = torch.empty(n, dtype=torch.uint8, pin_memory=True)
d_in = torch.empty(n, dtype=torch.uint8, device="cuda")
d_out = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(up, down):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if up:
        with torch.cuda.stream(s1):
            d_in.copy_(h_in, non_blocking=True)
    if down:
        with torch.cuda.stream(s2):
            h_out.copy_(d_out, non_blocking=True)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for _ in range(2):
    run(True, True)
for name, up, down in (("H2D", True, False), ("D2H", False, True), ("both", True, True)):
    t = min(run(up, down) for _ in range(3))
    gb = (up + down) * n / 1e9
    print(f"{name}: {gb / t:.1f} GB/s ({t * 1e3:.1f} ms for {gb:.2f} GB)")
