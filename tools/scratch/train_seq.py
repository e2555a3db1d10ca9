"""Kernel sequence of ONE rollout step of training's stepping pass from a rocprofv3 kernel trace (rocpd db): python tools/scratch/train_seq.py x.db [marker] [which]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "knn_multi_kernel"
which = int(sys.argv[3]) if len(sys.argv) > 3 else 130
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
t = lambda stem: next(x for x in tabs if x.startswith(stem))
kd, ks = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
name_col = "kernel_name" if "kernel_name" in [r[1] for r in db.execute(f"pragma table_info({ks})")] else "display_name"
rows = db.execute(f"select s.{name_col}, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
idx = [i for i, r in enumerate(rows) if marker in r[0]]
print(len(rows), "dispatches;", len(idx), "markers")
a, b = idx[which], idx[which + 1]
tot = 0
for n, s, e in rows[a:b]:
    tot += e - s
    print(f"{(s - rows[a][1]) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  {n[:110]}")
print("launches", b - a, "busy us", tot / 1e3, "span us", (rows[b][1] - rows[a][1]) / 1e3)
