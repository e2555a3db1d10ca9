#!/usr/bin/env python
"""Key figures of a bench.py JSON line (stdin or file): headline, per-shape values, training, cpu baseline."""
import json
import sys

txt = (open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin).read()
line = [l for l in txt.splitlines() if l.startswith("{") and '"metric"' in l][-1]
d = json.loads(line)


def shape(name, s):
    if not s:
        return
    r = s.get("roofline") or {}
    print(f"{name:10s} value {s['value']:12.1f}  ms/step {s['ms_per_step']:.4f}  all {['%.4f' % v for v in s.get('ms_per_step_all', [])]}"
          f"  end_to_end {s.get('end_to_end_value', 0):.0f}  roofline {r.get('kernel')} {r.get('avg_launch_us', 0):.1f} us frac {r.get('frac')}")


shape("headline", d)
cfg = d.get("config", {})
print("   ", cfg.get("workload"))
shape("c5", d.get("wosac_shape"))
b = d.get("bf16") or {}
shape("bf16 c2", b)
shape("bf16 c5", b.get("wosac_shape"))
t = d.get("training")
if t:
    print(f"training   {t['value']:.2f} scenes/s  {t['ms_per_step']:.1f} ms/step")
c = d.get("cpu_baseline")
if c:
    print(f"cpu        {c['value']:.0f} {c['unit']} on {c['cores']} threads; speedup {d.get('speedup_vs_cpu_baseline'):.1f}")
for k in d.get("kernels", []):
    print(f"    {k['class']:10s} {k['kernel'][:34]:34s} x{k['launches_per_step']:<4.0f} {k['avg_launch_us']:7.1f} us  share {k['share_of_step_kernel_time']:.3f}  frac {k.get('frac')}")
