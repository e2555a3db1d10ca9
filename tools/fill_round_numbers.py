"""Moves the round's figures in DESIGN.md / README.md from one judged line to another: every figure is formatted from the OLD line and from
the NEW line the same way and replaced as a string.   python tools/fill_round_numbers.py old_line.json new_line.json"""
import json, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def figures(l):
    k = lambda v: f"{v / 1e3:.0f} k"
    m = lambda v: f"{v / 1e6:.2f} M"
    b, t = l["bf16"], l["training"]
    r = l["roofline"]
    return {
        "head_checks": f"{k(l['with_rule_checks']['value'])} with the per-step",
        "head_checks2": f"| {k(l['with_rule_checks']['value'])} | |",
        "head": f"{k(l['value'])} ({l['ms_per_step']:.4f} ms/step)",
        "frac": f"{r['frac']:.3f} ({r['frac_at_trace_avg']:.3f} at the trace's {r['avg_launch_us_trace']:.1f} us)",
        "wosac": f"{m(l['wosac_shape']['value'])} ({l['wosac_shape']['ms_per_step']:.3f}); with rule checks {m(l['wosac_shape']['with_rule_checks']['value'])}",
        "sub": f"{m(l['submission_shape']['value'])} ({l['submission_shape']['ms_per_step']:.3f}); with rule checks + the 32-of-128 filter {m(l['submission_shape']['with_rule_checks']['value'])}",
        "batched": f"{m(l['batched']['value'])} ({l['batched']['ms_per_step']:.3f})",
        "bf16": f"{k(b['value'])} / {m(b['wosac_shape']['value'])} / {m(b['submission_shape']['value'])}",
        "train32": f"{l['training_fp32']['value']:.1f} ({l['training_fp32']['ms_per_step']:.1f} ms/step)",
        "train": f"{t['value']:.1f} scenes/s ({t['ms_per_step']:.1f} ms/step)",
        "cpu": f"{l['cpu_baseline']['value'] / 1e3:.2f} k ({l['speedup_vs_cpu_baseline']:.0f} x)",
    }


old, new = (figures(json.loads(Path(p).read_text())) for p in sys.argv[1:3])
for f in ("DESIGN.md", "README.md"):
    s = (ROOT / f).read_text()
    for key in sorted(old, key=lambda x: -len(old[x])):
        n = s.count(old[key])
        s = s.replace(old[key], new[key])
        print(f"{f}: {key}: {n} x '{old[key]}' -> '{new[key]}'")
    (ROOT / f).write_text(s)
