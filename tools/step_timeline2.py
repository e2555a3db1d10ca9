#!/usr/bin/env python
"""Kernel timeline of one steady-state step of the TWO-stream graph replay (bench.py default) from a rocprofv3 rocpd database:
every kernel between two consecutive agent-part tbx_sim_step launches, with its queue, start offset, duration and the idle gap
on its own queue before it.  usage: step_timeline2.py <db> [agents_per_step_blocks_min]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
t = lambda stem: next(x for x in tabs if x.startswith(stem))
kd, ks = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "tid")
rows = db.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x, d.{qcol} from {kd} d join {ks} s "
                  f"on d.kernel_id=s.id order by d.start").fetchall()
minb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
idx = [i for i, r in enumerate(rows) if "sim_step_kernel" in r[0] and r[3] // max(r[4], 1) >= minb]
if len(idx) >= 4:
    a, b = idx[-4], idx[-3]
    step = rows[a + 1:b + 1]
else:  # Schedule.fused_tail: the agents' step runs inside the last decoder layer's launch - a step = one searches' launch to the next
    idx = [i for i, r in enumerate(rows) if "knn_multi_kernel" in r[0]]
    if len(idx) < 4:  # Schedule.one_queue: no lights' tbx_sim_step either - a step = one tbx_front_pair launch to the next
        idx = [i for i, r in enumerate(rows) if "front_pair_kernel" in r[0]]
    a, b = idx[-4], idx[-3]
    step = rows[a:b]
t0 = step[0][1]
last_end = {}
busy = {}
for n, s, e, g, w, q in step:
    short = n.split("N_1")[-1][:40] if "GLOBAL" in n else n[:40]
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    busy[q] = busy.get(q, 0) + (e - s)
    print(f"q{q:<3} {(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  grid {g // max(w, 1):5d} x {w:4d}  {short}")
print("span", (step[-1][2] - t0) / 1e3, "us;", len(step), "launches; busy per queue:", {q: round(v / 1e3, 1) for q, v in busy.items()})
