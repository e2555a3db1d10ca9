#!/usr/bin/env python
"""Per-kernel summary (calls, total / avg / min / max duration) from a rocprofv3 rocpd SQLite database.
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/xx_kernel_stats.md"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    t = lambda stem: next(x for x in tabs if x.startswith(stem))
    kd, ks = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
    name_col = "kernel_name" if "kernel_name" in [r[1] for r in db.execute(f"pragma table_info({ks})")] else "display_name"
    q = (f"select s.{name_col}, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
         f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc")
    rows = db.execute(q).fetchall()
    tot = sum(r[2] for r in rows)
    print(f"| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|")
    for n, c, s, a, mn, mx in rows:
        n = n if len(n) < 90 else n[:87] + "..."
        print(f"| `{n}` | {c} | {s/1e6:.3f} | {a/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*s/tot:.1f} |")
    print(f"\ntotal kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches; dispatch columns: {cols}")


if __name__ == "__main__":
    main(sys.argv[1])
