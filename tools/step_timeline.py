#!/usr/bin/env python
"""Kernel timeline of one steady-state simulation step from a rocprofv3 rocpd database of `bench.py --no-graph`."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
t = lambda stem: next(x for x in tabs if x.startswith(stem))
kd, ks = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
rows = db.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from {kd} d join {ks} s "
                  f"on d.kernel_id=s.id order by d.start").fetchall()
idx = [i for i, r in enumerate(rows) if "sim_step_kernel" in r[0]]  # one per step in the one-stream order (--no-lights-ahead)
a, b = idx[-3], idx[-2]
step = rows[a + 1:b + 1]
t0, tot, agg = step[0][1], 0, {}
for n, s, e, g, w in step:
    short = n.split("N_1")[-1][:44] if "GLOBAL" in n else n[:44]
    if len(sys.argv) > 2:
        print(f"{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  grid {g // max(w, 1):5d} x {w:4d}  {short}")
    tot += e - s
    k = short.split("ENS")[0].split("EPK")[0]
    agg[k] = (agg.get(k, (0, 0))[0] + 1, agg.get(k, (0, 0))[1] + (e - s))
for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:50s} x{c:3d}  {d / 1e3:8.1f} us  ({100 * d / tot:4.1f} %)")
print("sum of kernel durations", tot / 1e3, "us over", len(step), "launches; span", (step[-1][2] - t0) / 1e3, "us")
