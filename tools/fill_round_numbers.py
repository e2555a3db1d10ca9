"""Fills the R6_* placeholders of DESIGN.md / README.md from the committed judged line (profiles/r06_bench_line.json).
   python tools/fill_round_numbers.py [tag]"""
import json, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
l = json.loads((ROOT / "profiles" / f"{tag}_bench_line.json").read_text())
k = lambda v: f"{v / 1e3:.0f} k"
m = lambda v: f"{v / 1e6:.2f} M"
b, t = l["bf16"], l["training"]
rep = {
    "R6_HEAD_CHECKS": k(l["with_rule_checks"]["value"]),
    "R6_HEAD": f"{k(l['value'])} ({l['ms_per_step']:.4f} ms/step)",
    "R6_FRAC": f"{l['roofline']['frac']:.3f}" + (f" ({l['roofline']['frac_at_trace_avg']:.3f} at the trace's {l['roofline']['avg_launch_us_trace']:.1f} us)" if l["roofline"].get("frac_at_trace_avg") else ""),
    "R6_WOSAC": f"{m(l['wosac_shape']['value'])} ({l['wosac_shape']['ms_per_step']:.3f}); with rule checks {m(l['wosac_shape']['with_rule_checks']['value'])}",
    "R6_SUB": f"{m(l['submission_shape']['value'])} ({l['submission_shape']['ms_per_step']:.3f}); with rule checks + the 32-of-128 filter {m(l['submission_shape']['with_rule_checks']['value'])}",
    "R6_BATCHED": f"{m(l['batched']['value'])} ({l['batched']['ms_per_step']:.3f})",
    "R6_BF16": f"{k(b['value'])} / {m(b['wosac_shape']['value'])} / {m(b['submission_shape']['value'])}",
    "R6_TRAIN32": f"{l['training_fp32']['value']:.1f} ({l['training_fp32']['ms_per_step']:.1f} ms/step)",
    "R6_TRAIN": f"{t['value']:.1f} scenes/s ({t['ms_per_step']:.1f} ms/step)",
    "R6_CPU": f"{l['cpu_baseline']['value'] / 1e3:.2f} k ({l['speedup_vs_cpu_baseline']:.0f} x)" if l.get("speedup_vs_cpu_baseline") else f"{l['cpu_baseline']['value'] / 1e3:.2f} k",
}
for f in ("DESIGN.md", "README.md"):
    s = (ROOT / f).read_text()
    for key in sorted(rep, key=len, reverse=True):
        s = s.replace(key, rep[key])
    (ROOT / f).write_text(s)
print(rep)
