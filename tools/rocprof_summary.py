#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (ROCm 7.2 default output) as text:
per-kernel calls / total / average duration (the `--stats` table) and, when the run
used --pmc, the per-kernel average of each counter.

    python tools/rocprof_summary.py gpurun_out/prof_stats/r01_results.db > profiles/r01_kernel_stats.txt
"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) "
                        "from kernels group by name order by sum(duration) desc"))
tot = sum(r[2] for r in rows)
print(f"# source: {sys.argv[1]}")
print(f"# total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
print(f"{'kernel':90s} {'calls':>6s} {'total_us':>11s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>7s} {'scr':>5s}")
for n, c, s, a, mn, mx, vg, ag, sg, lds, scr in rows:
    print(f"{n[:90]:90s} {c:6d} {s / 1e3:11.1f} {a / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * s / tot:6.2f} {vg:5d} {ag:5d} {sg:5d} {lds:7d} {scr:5d}")

try:
    cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
    if cols:
        print("# counters_collection columns:", cols)
        kcol = "kernel_name" if "kernel_name" in cols else "name"
        acc = defaultdict(lambda: [0.0, 0])
        for k, cn, v in cur.execute(f"select {kcol}, counter_name, value from counters_collection"):
            acc[(k, cn)][0] += v
            acc[(k, cn)][1] += 1
        if acc:
            print("\n# PMC counters: per-kernel average over dispatches (raw counter units)")
            for (k, cn), (s, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
                print(f"{k[:90]:90s} {cn:14s} avg={s / c:16.1f}  dispatches={c}")
except Exception as e:  # noqa
    print("# no counters:", e)
