#!/usr/bin/env python
"""Per-kernel average of one PMC counter from a rocprofv3 rocpd database (separate --pmc pass per counter).
    python tools/rocpd_pmc.py fetch.db write.db > profiles/rNN_pmc.json
FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3 counter definition). gfx950 correction (MI355X_MICROARCH.md §HBM):
FETCH_SIZE counts 128-B requests of wide coalesced reads at 64 B, i.e. reports half of the bytes of a 16 B/lane stream."""
import json
import sqlite3
import sys


def per_kernel(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    t = lambda stem: next(x for x in tabs if x.startswith(stem))
    pe, pi, kd, ks = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    q = (f"select s.kernel_name, p.name, count(*), avg(e.value), sum(e.value) from {pe} e join {pi} p on e.pmc_id = p.id "
         f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by s.kernel_name, p.name")
    out = {}
    for name, ctr, n, avg, tot in db.execute(q):
        short = name.split("(")[0]
        out.setdefault(short, {})[ctr] = {"launches": n, "avg": avg, "sum": tot}
    return out


def main(paths):
    res = {}
    for p in paths:
        for k, v in per_kernel(p).items():
            res.setdefault(k, {}).update(v)
    keep = {k: v for k, v in res.items() if "GLOBAL__N" in k}
    print(json.dumps(keep, indent=1))


if __name__ == "__main__":
    main(sys.argv[1:])
