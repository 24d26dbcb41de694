"""bench.py --gpus N and ranks (round-4 review, missing item 6): a plain `python bench.py --gpus N` starts N ranks by itself,
a launcher that started a different number of ranks is refused, and every reported line has n_gpus == --gpus.  CPU only: the
decision function and the command line it builds; the spawned ranks need GPUs."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("waa_bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_single_gpu_without_a_launcher_runs_in_process(bench):
    assert bench.launcher_decision(1, {}) == ("run",)


@pytest.mark.parametrize("n", [2, 4, 8])
def test_plain_invocation_with_n_gpus_spawns_n_ranks(bench, n):
    assert bench.launcher_decision(n, {}) == ("spawn",)


@pytest.mark.parametrize("n", [1, 2, 8])
def test_under_a_launcher_with_matching_world_size_runs(bench, n):
    assert bench.launcher_decision(n, {"WORLD_SIZE": str(n), "RANK": "0", "LOCAL_RANK": "0"}) == ("run",)


@pytest.mark.parametrize("gpus,world", [(8, 1), (1, 8), (4, 2)])
def test_a_mismatch_between_gpus_and_world_size_is_refused(bench, gpus, world):
    d = bench.launcher_decision(gpus, {"WORLD_SIZE": str(world)})
    assert d[0] == "error" and str(gpus) in d[1] and str(world) in d[1]


def test_bad_values_are_refused(bench):
    assert bench.launcher_decision(0, {})[0] == "error"
    assert bench.launcher_decision(2, {"WORLD_SIZE": "two"})[0] == "error"


def test_mismatch_exits_non_zero_before_touching_the_device():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_spawn_builds_the_torchrun_command_line(bench, monkeypatch):
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("MASTER_PORT", raising=False)
    assert bench.spawn_ranks(8, ["--gpus", "8", "--steps", "5", "--warmup", "1"]) == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 0 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" or os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") is not None
