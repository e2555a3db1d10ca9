"""The ONE JSON line the driver parses, and the detail side file.

The judged line is compact by construction (< 8 KB; tests/test_bench_line.py holds it to that): the contract's top-level keys,
`config`, `roofline`, `cpu_baseline`, and one small object each for the WOSAC shape, the bf16 tables and the training step. The
per-kernel arrays, every repeat, the notes and the scenes-per-GPU curve go to the detail file (default
gpurun_out/bench_detail.json) and, as a short table, to stderr."""
import json
import math
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
MAX_LINE_BYTES = 8192

ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "traffic_measured", "traffic_matches_build",
             "hbm_measured_frac", "avg_launch_us", "avg_launch_us_trace", "trace_source", "frac_at_trace_avg", "valu_issue_frac", "valu_issue_floor_us",
             "bound_8d", "launches_per_step", "share_of_step_kernel_time", "share_of_step", "algorithmic_bytes_per_launch",
             "compulsory_bytes_per_launch", "flops_per_launch", "bytes_per_pair", "source_rows_per_launch")


def _r(x, sig=6):
    """floats to `sig` significant digits (the line is for reading; the detail file keeps everything)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        if not math.isfinite(x):
            return None  # NaN / Infinity are not JSON
        if x == 0.0:
            return 0.0
        return round(x, sig - 1 - int(math.floor(math.log10(abs(x)))))
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


SUB_ROOF_KEYS = ("kernel", "bound", "bound_8d", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "traffic_measured", "hbm_measured_frac",
                 "valu_issue_frac", "valu_issue_floor_us", "avg_launch_us", "avg_launch_us_trace", "launches_per_step",
                 "share_of_step_kernel_time", "share_of_step", "algorithmic_bytes_per_launch", "flops_per_launch")


def _gathered_note(roof):
    """SURVEY 8d prices an attention launch by the bytes its pairs GATHER; rows shared by many source rows come from L2, so that figure
    over the HBM peak is not bounded by 1: say so where it happens."""
    if roof.get("bound") == "hbm" and (roof.get("frac") or 0) > 1.0:
        roof["frac_note"] = "8d gathered bytes (L2-served rows included) / s over HBM peak: not bounded by 1"


def compact_roofline(r, sub=False):
    """sub: the roofline of an appended shape (the judged object keeps every key; the detail file keeps everything of all of them)."""
    if not r:
        return None
    if "error" in r:
        return {"error": str(r["error"])[:200]}
    out = {k: r[k] for k in (SUB_ROOF_KEYS if sub else ROOF_KEYS) if k in r}
    out.setdefault("traffic", None)
    _gathered_note(out)
    c = r.get("counters")
    if c:
        out["counters"] = {k: c[k] for k in ("valu_busy", "l2_hit_rate", "valu_insts_per_pair", "counters_source") if k in c}
    return out


def compact_shape(res, keys=("value", "ms_per_step", "ms_per_step_min", "steps", "warmup", "end_to_end_value", "finite"), lean=False):
    """lean: a nested shape (the WOSAC shape inside the bf16 object): value, ms/step and the checked value only."""
    if not res:
        return None
    if "error" in res and "value" not in res:
        return {"error": str(res["error"])[:200]}
    out = {k: res[k] for k in keys if k in res}
    if lean:
        w = res.get("with_rule_checks") or {}
        if "value" in w:
            out["with_rule_checks"] = {"value": w["value"], "vs_unchecked": w.get("vs_unchecked")}
        r = res.get("roofline") or {}
        if "frac" in r:  # (frac = algorithmic bytes or flops per launch / avg_launch_us / peak, as everywhere; the full object is in the detail file)
            out["roofline"] = {k: r[k] for k in ("kernel", "bound", "frac", "avg_launch_us", "algorithmic_bytes_per_launch", "flops_per_launch", "peak", "traffic", "traffic_source") if r.get(k) is not None}
            _gathered_note(out["roofline"])
        return out
    if res.get("scene_reuse"):
        out["new_scene_ms"] = res["scene_reuse"]["new_scene_ms"]
    if res.get("config"):
        wl = res["config"].get("workload") or ""
        out["workload"] = wl.split(" synthetic")[0] if " synthetic" in wl else wl[:120]  # (the long form is the headline's config.workload)
        for k in ("scenes_per_gpu", "rollouts_per_scene", "global_batch", "parallelism", "allreduce_bytes"):
            if k in res["config"]:
                out[k] = res["config"][k]
    if "dtype" in res:
        out["dtype"] = res["dtype"]
    out["roofline"] = compact_roofline(res.get("roofline"), sub=True)
    if res.get("with_rule_checks"):
        out["with_rule_checks"] = compact_checks(res["with_rule_checks"])
    return out


def compact_checks(w):
    if "error" in w:
        return {"error": str(w["error"])[:200]}
    return {k: w[k] for k in ("value", "ms_per_step", "rule_checks_ms", "vs_unchecked", "filter_futures") if k in w}


def judged_line(full):
    """full: bench.py's complete result (headline fields + wosac_shape / bf16 / training / cpu_baseline sub-results)."""
    head_keys = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data")
    line = {k: full.get(k) for k in head_keys}
    cfg = dict(full.get("config") or {})
    cfg.pop("weights", None)
    line["config"] = cfg
    line["roofline"] = compact_roofline(full.get("roofline"))
    if full.get("roofline_gemm"):
        g = full["roofline_gemm"]
        line["roofline_gemm"] = {k: g[k] for k in ("bound", "achieved", "peak", "unit", "frac") if k in g}
    for k in ("ms_per_step_min", "repeats", "end_to_end_value", "first_scene_value", "scene_encode_ms", "graph_capture_ms", "finite"):
        if k in full:
            line[k] = full[k]
    if full.get("scene_reuse"):
        line["new_scene_ms"] = full["scene_reuse"]["new_scene_ms"]
    if full.get("with_rule_checks"):
        line["with_rule_checks"] = compact_checks(full["with_rule_checks"])
    if "cpu_baseline" in full:
        line["cpu_baseline"] = full["cpu_baseline"]
        line["speedup_vs_cpu_baseline"] = full.get("speedup_vs_cpu_baseline")
    if full.get("wosac_shape"):
        line["wosac_shape"] = compact_shape(full["wosac_shape"])
    for k in ("submission_shape", "batched"):
        if full.get(k):
            line[k] = compact_shape(full[k], keys=("value", "ms_per_step", "ms_per_step_min", "steps", "warmup", "finite"))
    if full.get("bf16"):
        b = compact_shape(full["bf16"])
        if full["bf16"].get("wosac_shape"):
            b["wosac_shape"] = compact_shape(full["bf16"]["wosac_shape"], keys=("value", "ms_per_step", "steps", "warmup", "finite"), lean=True)
        if full["bf16"].get("submission_shape"):
            b["submission_shape"] = compact_shape(full["bf16"]["submission_shape"], keys=("value", "ms_per_step", "steps", "warmup", "finite"), lean=True)
        line["bf16"] = b
    if full.get("reduced"):
        r = compact_shape(full["reduced"])
        if full["reduced"].get("wosac_shape"):
            r["wosac_shape"] = compact_shape(full["reduced"]["wosac_shape"], keys=("value", "ms_per_step", "steps", "warmup", "finite"), lean=True)
        line["reduced"] = r
    if full.get("training"):
        t = compact_shape(full["training"], keys=("metric", "value", "unit", "ms_per_step", "steps", "warmup", "loss", "finite", "dtype"))
        t["workload"] = "training_step fwd+bwd+grad all-reduce+AdamW, 16 scenes/GPU (64/1024/128), 90-step rollout" if t.get("workload") else None
        t["train_precision"] = (full["training"].get("config") or {}).get("train_precision")
        line["training"] = t
    if full.get("training_fp32"):
        f32 = full["training_fp32"]
        line["training_fp32"] = {"error": str(f32["error"])[:200]} if "error" in f32 else {k: f32[k] for k in ("value", "ms_per_step", "steps", "warmup", "loss", "finite") if k in f32}
    if full.get("scene_curve"):
        line["scene_curve"] = [{"scenes": c["scenes"], "value": c["value"], "ms_per_step": c["ms_per_step"]} for c in full["scene_curve"]]
    if full.get("detail_file"):
        line["detail_file"] = full["detail_file"]
    return _r(line)


def shrink(line):
    """Last resort if a line still exceeds the limit (long error strings, a long curve): drop optional objects, largest first."""
    for k in ("scene_curve", "reduced", "training_fp32", "submission_shape", "batched", "bf16", "roofline_gemm", "wosac_shape", "training"):
        if len(json.dumps(line)) < MAX_LINE_BYTES:
            break
        if k in line:
            line[k] = {"omitted": "see detail_file"}
    return line


def write_detail(full, path):
    """Everything measured, as one indented JSON document. Returns the path written (relative to the repo), or None."""
    if path == "-":
        return None
    p = Path(path) if path else ROOT / "gpurun_out" / "bench_detail.json"
    try:
        p.parent.mkdir(parents=True, exist_ok=True)
        with open(p, "w") as f:
            json.dump(_r(full, 8), f, indent=1)
        try:
            return str(p.resolve().relative_to(ROOT))
        except ValueError:
            return str(p)
    except OSError as e:  # a read-only checkout: the judged line does not depend on the side file
        print(f"bench.py: detail file not written ({e})", file=sys.stderr)
        return None


def kernel_table(name, res, out=sys.stderr):
    ks = (res or {}).get("kernels")
    if not ks:
        return
    print(f"-- {name}: {res.get('value', 0):.1f} {res.get('unit', '')} at {res.get('ms_per_step', 0):.4f} ms/step", file=out)
    for k in ks:
        share = k.get("share_of_step_kernel_time", k.get("share_of_step"))
        avg = k.get("avg_launch_us")
        print(f"   {str(k.get('kernel'))[:56]:56s} x{k.get('launches_per_step', 0):<6.1f} {0.0 if avg is None else avg:8.1f} us  share {0.0 if share is None else share:.3f}"
              f"  {k.get('bound')} frac {k.get('frac')}", file=out)


def emit(full, detail_path=None, out=sys.stdout):
    """Writes the detail file, prints the per-kernel tables to stderr and the judged line (ONE line, last on stdout)."""
    full = dict(full)
    full["detail_file"] = write_detail(full, detail_path)
    kernel_table("headline", full)
    kernel_table("wosac_shape", full.get("wosac_shape"))
    kernel_table("submission_shape", full.get("submission_shape"))
    kernel_table("batched", full.get("batched"))
    kernel_table("training", full.get("training"))
    line = shrink(judged_line(full))
    s = json.dumps(line, allow_nan=False)
    assert "\n" not in s
    sys.stderr.flush()
    print(s, file=out, flush=True)
    return s
