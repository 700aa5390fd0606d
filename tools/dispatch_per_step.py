#!/usr/bin/env python
"""Steady-state dispatches per step from TWO rocprofv3 kernel traces of the same command with different step counts
(one-time work -- weight packing, transposes, first-call caches -- cancels):
    python tools/dispatch_per_step.py short.db long.db <extra steps in long>
-> per kernel: dispatches per step and us per step; the non-gn:: (framework) launches listed separately."""
import sqlite3
import sys
from collections import defaultdict


def table(path):
    cur = sqlite3.connect(path).cursor()
    t = defaultdict(lambda: [0, 0.0])
    for n, c, s in cur.execute("select name, count(*), sum(duration) from kernels group by name"):
        t[n] = [c, s / 1e3]
    return t


a, b, extra = table(sys.argv[1]), table(sys.argv[2]), float(sys.argv[3])
rows = []
for k in set(a) | set(b):
    dc, dt = (b[k][0] - a[k][0]) / extra, (b[k][1] - a[k][1]) / extra
    if abs(dc) > 1e-9:
        rows.append((k, dc, dt))
rows.sort(key=lambda r: -r[2])
gn = [r for r in rows if "gn::" in r[0]]
fw = [r for r in rows if "gn::" not in r[0]]
print(f"# steady-state dispatches per step = (long - short) / {extra:g} steps  ({sys.argv[1]} vs {sys.argv[2]})")
print(f"# gn:: kernels: {sum(r[1] for r in gn):.1f} dispatches, {sum(r[2] for r in gn):.1f} us per step;  "
      f"framework (non-gn::) launches: {sum(r[1] for r in fw):.1f} dispatches, {sum(r[2] for r in fw):.1f} us per step")
print("# ---- framework launches per step")
for k, dc, dt in fw:
    print(f"{k[:110]:110s} {dc:7.2f} {dt:9.2f} us")
print("# ---- gn:: kernels per step")
for k, dc, dt in gn:
    print(f"{k[:110]:110s} {dc:7.2f} {dt:9.2f} us")
