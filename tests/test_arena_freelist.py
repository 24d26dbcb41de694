"""The device arena's bookkeeping (csrc/waa_freelist.hpp: first-fit free list with coalescing; ADVICE round 4 — the bump
allocator it replaces never returned a byte while batch lifetimes overlapped) against a byte-map model, compiled with g++ and
run on the CPU: no overlap, alignment, legal misses only, full coalescing, the pipeline pattern never misses."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("freelist") / "freelist_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tools", "freelist_check.cpp")])
    return exe


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_free_list_against_the_model(checker, seed):
    r = subprocess.run([checker, str(seed)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout
    assert int(r.stdout.split()[2]) > 0  # the sequence did run into legal misses (the slab is small on purpose)
