#!/usr/bin/env python
"""Turn rocprofv3 result databases (gpurun_out/.../*_results.db) into the small text summaries
committed under profiles/.   usage: rocprof_summary.py <results.db> [<results.db> ...]"""
import sqlite3
import sys


def short(name):
    return name if len(name) < 90 else name[:87] + "..."


for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    cur = db.cursor()
    print(f"== {path}")
    print("-- kernel stats (rocprofv3 --kernel-trace --stats): name, calls, total_us, avg_us, pct")
    for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 12"):
        print(f"{short(r[0])} | {r[1]} | {r[2]:.1f} | {r[3]:.3f} | {r[4]:.2f}")
    try:
        cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
        rows = list(cur.execute("select * from counters_collection"))
    except sqlite3.Error:
        rows = []
    if rows:
        ki, ci, vi = cols.index("kernel_name"), cols.index("counter_name"), cols.index("value")
        agg = {}
        for r in rows:
            agg.setdefault((r[ki], r[ci]), []).append(float(r[vi]))
        print("-- PMC per dispatch: kernel, counter, dispatches, mean value")
        for (k, c), v in sorted(agg.items()):
            print(f"{short(k)} | {c} | {len(v)} | {sum(v) / len(v):.1f}")
