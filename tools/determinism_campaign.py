#!/usr/bin/env python
"""tools/determinism_campaign.py --first F --count N --repeats K --jobs J [--hot SEED,...] [--out FILE] — bitwise determinism
of the HIP library's renders (GPU box).

Every random graph of both generators of tests/test_fuzz_graphs.py (seeds [F, F + N)) is built ONCE and rendered K times
in the same process (waa_render re-arms every recurrence, delay line and convolver window); renders 2..K are compared BIT FOR
BIT (as u32 words, so NaNs compare too) with render 1.  Every `--rebuild-every`-th graph is in addition built a second time from
scratch (fresh allocations, fresh plan-time uploads and plan-time kernels) and that render compared with render 1 as well.
No oracle is involved: this campaign asks one question — does the SAME process give the SAME bits twice — and answers it for
the deployment model (ONE process per device, --jobs 1) and for the oversubscribed model the parity campaigns use (--jobs 8:
eight processes time-sliced on one device), on the same seeds.  `--hot` names seeds to hammer (rendered `--hot-repeats` times):
the four non-repeating sightings of round 3 are the default.

The record (JSON) holds per job count: graphs, renders, bit-identical renders, refused graphs (status 4), every differing render
(seed, generator, which repeat, where: instance / channel / count / first / last frame, max |diff|, whether a fresh rebuild
repeats it) and the wall time; plus the library build it ran on."""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HOT_DEFAULT = "0:308957,0:212240,0:606992,0:200667,1:308957,1:212240,1:606992,1:200667"


def _download(ch, out):
    import ctypes as C
    return ch._b.download_all(ch._handle, out.ctypes.data_as(C.POINTER(C.c_float)))


def worker(first, count, frozen, repeats, rebuild_every, hot, hot_repeats, out_path, seconds):
    import numpy as np

    import web_audio_api_rs_amd as waa
    from test_fuzz_graphs import build_random_graph

    waa.set_hrtf_database(os.path.join(ROOT, "tests", "golden", "IRC_1003_C.bin"))
    hip = waa.default_binding()
    rec = {"graphs": 0, "renders": 0, "identical_renders": 0, "rebuilds": 0, "identical_rebuilds": 0, "refused": 0,
           "different": [], "errors": [], "seconds": 0.0, "stopped_early_at_seed": None}
    t0 = time.time()

    def describe(a, b):
        d = a.view(np.uint32) != b.view(np.uint32)
        where = []
        for i in range(a.shape[0]):
            for c in range(a.shape[1]):
                bad = np.nonzero(d[i, c])[0]
                if len(bad):
                    where.append([i, c, int(len(bad)), int(bad[0]), int(bad[-1])])
        with np.errstate(invalid="ignore"):
            mx = float(np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64))))
        return where[:8], mx

    def one(seed, fz, k_renders, rebuild):
        ch, descr = build_random_graph(hip, seed, frozen=fz)
        try:
            try:
                ch.render_async()
            except waa.WaaError as e:
                if e.status == 4:
                    rec["refused"] += 1
                    return
                raise
            base = np.empty((ch.n_instances, ch.number_of_channels, ch.length), np.float32)
            ch._b.check(_download(ch, base))
            rec["graphs"] += 1
            rec["renders"] += 1
            for r in range(1, k_renders):
                ch.render_async()
                again = np.empty_like(base)
                ch._b.check(_download(ch, again))
                rec["renders"] += 1
                if np.array_equal(base.view(np.uint32), again.view(np.uint32)):
                    rec["identical_renders"] += 1
                    continue
                where, mx = describe(base, again)
                # which of the two is the odd one: a third render, and a context built from scratch
                ch.render_async()
                third = np.empty_like(base)
                ch._b.check(_download(ch, third))
                ch2, _ = build_random_graph(hip, seed, frozen=fz)
                fresh = ch2.start_rendering_sync().data
                ch2.close()
                rec["different"].append({
                    "seed": seed, "frozen_generator": bool(fz), "repeat": r, "graph": str(descr)[:300],
                    "where_inst_ch_n_first_last": where, "max_abs_diff": mx,
                    "third_render_equals": "first" if np.array_equal(third.view(np.uint32), base.view(np.uint32)) else
                                           "differing" if np.array_equal(third.view(np.uint32), again.view(np.uint32)) else "neither",
                    "fresh_build_equals": "first" if np.array_equal(fresh.view(np.uint32), base.view(np.uint32)) else
                                          "differing" if np.array_equal(fresh.view(np.uint32), again.view(np.uint32)) else "neither"})
            if rebuild:
                ch2, _ = build_random_graph(hip, seed, frozen=fz)
                fresh = ch2.start_rendering_sync().data
                ch2.close()
                rec["rebuilds"] += 1
                rec["renders"] += 1
                if np.array_equal(fresh.view(np.uint32), base.view(np.uint32)):
                    rec["identical_rebuilds"] += 1
                else:
                    where, mx = describe(base, fresh)
                    rec["different"].append({"seed": seed, "frozen_generator": bool(fz), "repeat": "rebuild",
                                             "graph": str(descr)[:300], "where_inst_ch_n_first_last": where, "max_abs_diff": mx})
        finally:
            ch.close()

    for fz, seed in hot:
        try:
            one(seed, fz, hot_repeats, True)
        except Exception as e:  # noqa: BLE001
            rec["errors"].append({"seed": seed, "error": repr(e)[:300]})
    for seed in range(first, first + count):
        if seconds and time.time() - t0 > seconds:
            rec["stopped_early_at_seed"] = seed
            break
        try:
            one(seed, frozen, repeats, rebuild_every > 0 and seed % rebuild_every == 0)
        except Exception as e:  # noqa: BLE001 — a crash of one seed is a finding, not the end of the run
            rec["errors"].append({"seed": seed, "error": repr(e)[:300]})
    rec["seconds"] = round(time.time() - t0, 1)
    json.dump(rec, open(out_path, "w"))


def run(first, count, repeats, jobs, rebuild_every, hot, hot_repeats, out, seconds):
    """J processes per generator would double the load of the parity campaigns: here the J jobs split BOTH generators' seeds
    between them (job j renders seeds first + j, first + j + J, ... as contiguous ranges of the plain generator, then of the
    frozen-state one), so --jobs 1 really is ONE process on the device."""
    t0 = time.time()
    procs = []
    per = (count + jobs - 1) // jobs
    for j in range(jobs):
        lo = first + j * per
        n = max(0, min(per, first + count - lo))
        for frozen in (0, 1):
            part = f"{out}.part{jobs}_{frozen}_{j}"
            hot_arg = hot if (j == 0 and frozen == 0) else ""
            procs.append((part, [sys.executable, os.path.abspath(__file__), "--worker", str(lo), str(n), str(frozen), str(repeats),
                                 str(rebuild_every), hot_arg, str(hot_repeats), part, str(seconds / 2 if seconds else 0)]))
    # one process per JOB: the two generators of a job run one after the other inside that job's slot
    running = []
    for j in range(jobs):
        mine = [p for k, p in enumerate(procs) if k // 2 == j]
        running.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--chain"] +
                                        [json.dumps(cmd) for _, cmd in mine]))
    for p in running:
        p.wait()
    total = {"jobs": jobs, "graphs": 0, "renders": 0, "identical_renders": 0, "rebuilds": 0, "identical_rebuilds": 0,
             "refused": 0, "different": [], "errors": [], "stopped_early": []}
    for part, _ in procs:
        if not os.path.exists(part):
            total["errors"].append({"part": os.path.basename(part), "error": "worker wrote no record"})
            continue
        r = json.load(open(part))
        os.remove(part)
        for k in ("graphs", "renders", "identical_renders", "rebuilds", "identical_rebuilds", "refused"):
            total[k] += r[k]
        total["different"] += r["different"]
        total["errors"] += r["errors"]
        if r["stopped_early_at_seed"] is not None:
            total["stopped_early"].append(r["stopped_early_at_seed"])
    total["wall_s"] = round(time.time() - t0, 1)
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=800000)
    ap.add_argument("--count", type=int, default=25000, help="seeds per generator")
    ap.add_argument("--repeats", type=int, default=6, help="renders per built graph")
    ap.add_argument("--rebuild-every", type=int, default=8)
    ap.add_argument("--jobs", default="1,8", help="comma-separated process counts, run one after the other on the same seeds")
    ap.add_argument("--hot", default=HOT_DEFAULT, help="generator:seed,... rendered --hot-repeats times each (by job 0)")
    ap.add_argument("--hot-repeats", type=int, default=300)
    ap.add_argument("--seconds", type=float, default=0.0, help="time budget per job count (0 = none)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r04_determinism.json"))
    ap.add_argument("--worker", nargs=9, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--chain", nargs="+", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.chain:
        for cmd in args.chain:
            subprocess.call(json.loads(cmd))
        return
    if args.worker:
        w = args.worker
        hot = [(int(x.split(":")[0]), int(x.split(":")[1])) for x in w[5].split(",") if x]
        worker(int(w[0]), int(w[1]), int(w[2]), int(w[3]), int(w[4]), hot, int(w[6]), w[7], float(w[8]))
        return
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    lib = os.path.join(ROOT, "web-audio-api-rs_amd", "csrc", "libwaa_hip.so")
    out = {"what": "bitwise determinism of repeated renders: random Web Audio graphs (tests/test_fuzz_graphs.py::build_random_graph, "
                   "both generators, 3 contexts x 10440 frames), each built once and rendered `repeats` times in one process; "
                   "renders 2.. compared as u32 words with render 1; every `rebuild_every`-th graph also rebuilt from scratch",
           "seeds_per_generator": [args.first, args.first + args.count], "repeats": args.repeats,
           "rebuild_every": args.rebuild_every, "hot_seeds": args.hot, "hot_repeats": args.hot_repeats,
           "libwaa_hip_sha16": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16], "runs": []}
    for jobs in [int(x) for x in args.jobs.split(",")]:
        out["runs"].append(run(args.first, args.count, args.repeats, jobs, args.rebuild_every, args.hot, args.hot_repeats,
                               args.out, args.seconds))
        json.dump(out, open(args.out, "w"), indent=1)
        r = out["runs"][-1]
        print(json.dumps({k: (v if k != "different" else len(v)) for k, v in r.items()}), flush=True)


if __name__ == "__main__":
    main()
