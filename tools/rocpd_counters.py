#!/usr/bin/env python
"""Per-kernel counter sums of one rocprofv3 --pmc pass (rocpd database): for every tbx kernel the per-launch SUM over the samples
(shader engines x XCDs) of each counter, the launch count and the average launch duration.
    python tools/rocpd_counters.py pass.db [more.db ...] > out.json"""
import json
import re
import sqlite3
import sys


def short(mangled: str) -> str:
    m = re.search(r"GLOBAL__N_1\d+([a-z_0-9]+?_kernel)(I(?:L[ib]\d+E)+E)?", mangled)
    if not m:
        return mangled[:60]
    targs = re.findall(r"L[ib](\d+)E", m.group(2) or "")
    return m.group(1) + ("<" + ",".join(targs) + ">" if targs else "")


out = {}
for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    t = lambda stem: next(x for x in tabs if x.startswith(stem))
    pe, pi, kd, ks = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    q = (f"select s.kernel_name, p.name, sum(e.value), count(distinct d.id), avg(d.end - d.start) from {pe} e join {pi} p on e.pmc_id = p.id "
         f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by s.kernel_name, p.name")
    for name, ctr, total, n, dur in db.execute(q):
        if "GLOBAL__N" not in name or "at6native" in name:
            continue
        e = out.setdefault(short(name), {"launches": n, "avg_launch_ns": round(dur)})
        e[ctr] = total / max(n, 1)
for k, e in out.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "SQ_BUSY_CYCLES" in e and e["SQ_BUSY_CYCLES"] > 0:
        # SQ_BUSY_CYCLES: summed over the shader engines that were busy; MFMA-busy cycles are per SIMD, summed: /(4 SIMDs x 256 CUs) gives
        # the average SIMD's busy cycles, against the launch's cycles (GRBM_GUI_ACTIVE per sample when collected, else duration x 2.4 GHz)
        cyc = e.get("GRBM_GUI_ACTIVE_per_sample") or e["avg_launch_ns"] * 2.4
        e["mfma_busy_frac_of_launch"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc
print(json.dumps(dict(sorted(out.items(), key=lambda kv: -kv[1]["launches"] * kv[1]["avg_launch_ns"])), indent=1))
