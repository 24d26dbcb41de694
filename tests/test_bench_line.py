"""The bench line the round's final build printed under the driver's protocol (profiles/r06z_bench_default.json) against the contract of the task statement:
one JSON object with the fixed keys, BASELINE.json's metric, the `roofline` and `cpu_baseline` objects — and the riders the
driver's record keeps only the tail of (the north-star graph's record and what one start_rendering_sync costs) at the END of
the line.  CPU only: it reads a committed artefact; bench.py itself needs a GPU."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = os.path.join(ROOT, "profiles", "r06z_bench_default.json")


@pytest.fixture(scope="module")
def line():
    if not os.path.exists(LINE):
        pytest.skip("no committed bench line")
    text = open(LINE).read().strip()
    assert "\n" not in text  # ONE line
    return text, json.loads(text)


def test_contract_keys(line):
    _, d = line
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] in base["metric"] and d["unit"] == "quanta/s"
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    # value = contexts x quanta per context / seconds per step
    cfg = d["config"]
    assert abs(d["value"] - cfg["contexts_per_gpu"] * cfg["quanta_per_context"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]


def test_roofline_and_cpu_baseline_objects(line):
    _, d = line
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # algorithmic bytes of one launch: SURVEY 8(d), 2048 B x contexts x quanta
    cfg = d["config"]
    assert r["algorithmic_bytes_per_launch"] == 2048.0 * cfg["contexts_per_gpu"] * cfg["quanta_per_context"]
    assert r["traffic"] and 0.98 <= r["traffic"] / r["algorithmic_bytes_per_launch"] <= 1.03   # measured in the run: no wasted re-reads
    assert "measured in this run" in r["traffic_source"]
    # round 6: no `box_copy_floor` any more (a fresh process's probe measured the region IT was handed: product kernels beat it);
    # the like-for-like reference is the graded arena's own best unit — the same copy shape, same run — which the kernel cannot beat
    assert "box_copy_floor" not in r
    a = d["arena"]
    assert a["GiB"] > 0 and a["candidates_GiB"] >= a["GiB"] and a["misses"] == 0
    g = a["copy_into_unit_ms"]
    assert 0 < g["best"] <= g["worst_kept"] <= g["worst_candidate"]
    assert 0.8 <= r["kernel_over_best_region_copy"] <= 1.0
    assert abs(a["best_unit_copy_GBps"] - 2 * a["unit_GiB"] * 2**30 / (g["best"] * 1e-3) / 1e9) < 1.0
    # ... and the headline once more on plain hipMalloc, first batch of the process (rounds 1-5's protocol), in the same line
    cold = d["cold"]
    assert cold["kernel_ms"] > 0 and 0.4 < cold["frac"] < 0.9
    assert r["frac"] >= 0.68  # (round-5 review, item 1)
    assert r["kernel_ms"] <= d["configs"]["echo"]["kernel_ms"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "quanta/s" and c["sample"]


def test_the_riders_the_driver_keeps_are_at_the_end(line):
    text, d = line
    keys = list(d.keys())
    assert keys[-2:] == ["t1", "one_shot"], keys[-4:]
    tail = text[-2000:]
    assert '"t1":' in tail and '"kernel_frac"' in tail and '"one_shot"' in tail
    t1 = d["t1"]
    assert t1["traffic_source"].startswith("measured in this run") and set(t1["kernel_frac"]) == {"conv_mac_kernel", "conv_fft_kernel<inv>", "conv_fft_kernel<fwd>"}
    assert t1["ms"] < 8.3  # (round 4: 8.2-8.5)
    assert "sample" in t1["cpu_baseline"]  # (round-5 review, weak 9: the ratio is reproducible from the record)
    assert d["e2e"]["t1_ms"] > 0 and d["one_shot"]["create_ms"] >= 0
    for k in ("c3", "c4", "c5"):
        assert d["configs"][k].get("traffic_live") is True
