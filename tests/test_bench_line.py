"""bench.py's judged line (CPU): compact (< 8 KB on ONE line, json round trip, no NaN), carrying the contract's keys + `roofline` +
`cpu_baseline`, whatever the size of the full measurement behind it; and `python bench.py --gpus N` without a launcher starts its
N ranks itself (checked with --dry-run: gloo rendezvous, no GPU)."""
import io
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from tools.benchlib import launch, report  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline")


def _full_r03():
    """Round 3's complete line (30 KB: the one the driver could not parse) as a realistic full measurement."""
    return json.load(open(ROOT / "profiles" / "r03_bench_line.json"))


def test_judged_line_is_compact_and_complete():
    full = _full_r03()
    assert len(json.dumps(full)) > 20000  # (the input really is the oversized one)
    full["scene_curve"] = [{"scenes": s, "value": 1e5 * s, "ms_per_step": 0.6, "ms_per_step_min": 0.59, "finite": True} for s in (1, 2, 4, 16, 64)]
    out = io.StringIO()
    s = report.emit(full, "-", out=out)
    assert out.getvalue() == s + "\n" and "\n" not in s
    assert len(s.encode()) < report.MAX_LINE_BYTES, len(s)
    assert len(s.encode()) < 6144, len(s)  # headroom: today's line is ~4 KB
    line = json.loads(s)
    for k in CONTRACT:
        assert k in line, k
    r = line["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 * r["frac"]
    c = line["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(c) and c["kind"] in ("port", "reference")
    assert "workload" in line["config"] and "model" not in line["config"]
    for sub in ("wosac_shape", "bf16", "training"):
        assert "value" in line[sub] and "ms_per_step" in line[sub] and "kernels" not in line[sub]
        assert "roofline" in line[sub]
    assert "kernels" not in line and "ms_per_step_all" not in line
    assert abs(line["value"] - full["value"]) < 1e-5 * full["value"]  # (rounded to 6 significant digits, not changed)
    assert [c["scenes"] for c in line["scene_curve"]] == [1, 2, 4, 16, 64]


def test_judged_line_survives_nan_and_long_errors():
    full = _full_r03()
    full["training"] = {"error": "RuntimeError: " + "x" * 50000}
    full["roofline"]["achieved"] = float("nan")
    full["wosac_shape"]["roofline"] = {"error": "y" * 20000}
    s = report.emit(full, "-", out=io.StringIO())
    assert len(s.encode()) < report.MAX_LINE_BYTES
    line = json.loads(s)  # strict JSON: no NaN / Infinity tokens
    assert line["roofline"]["achieved"] is None and "NaN" not in s and len(line["training"]["error"]) <= 200


def test_detail_file_keeps_the_kernel_arrays(tmp_path):
    full = _full_r03()
    p = tmp_path / "detail.json"
    s = report.emit(full, str(p), out=io.StringIO())
    d = json.load(open(p))
    assert len(d["kernels"]) == len(full["kernels"]) and len(d["training"]["kernels"]) == len(full["training"]["kernels"])
    assert json.loads(s)["detail_file"] == str(p)


def test_gpus_n_without_a_launcher_spawns_its_ranks():
    assert launch.needs_spawn(1) is False
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = launch.launcher_command("bench.py", 8, ["--gpus", "8"], 12345)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-run", "--scenes", "3"], env=env, cwd=str(ROOT),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["max_rank_seen"] == 1 and line["config"]["scene_ids_rank0"] == [0, 1, 2]
    # a launcher's world that disagrees with --gpus is refused with a non-zero code, not an assertion trace
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-run"], env={**env, "WORLD_SIZE": "1", "RANK": "0"},
                       cwd=str(ROOT), capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr


# ---- the judged roofline follows SURVEY 8d and this round's counters (VERDICT r04 "what's weak" 3)
def _args(**kw):
    import argparse

    return argparse.Namespace(**{**dict(agents=64, polylines=1024, lights=128, scenes=1, rollouts=1, kv_bf16=False, attn_mfma=None), **kw})


def test_dec_layer_roofline_is_survey_8d_bytes_over_event_time():
    """`algorithmic_bytes_per_launch` of the one-launch decoder layer = the two attention calls' SURVEY 8d bytes and nothing else
    (S*2*d*b + P*(2*d*b + 17) + (d_rpe*2d + 2d)*b per call); weight images and token rows are reported beside it
    (`compulsory_bytes_per_launch`); frac = those bytes / the average event time / 8 TB/s."""
    from tools.benchlib import events

    rows, pairs_per_row = 64, 25 + 64 + 25  # configs[1]'s agents: K_aa + K_am + K_at
    b8d = rows * pairs_per_row * 1041 + 2 * (rows * 2 * 128 * 4) + 2 * (128 * 256 + 256) * 4
    assert b8d == 2 * events.attn_algorithmic_bytes(rows, 0, 4) + rows * pairs_per_row * (2 * 128 * 4 + 17) == 7990400
    c = dict(cls="dec_layer", key=rows, t=4 * 25e-6, n=4, work=4.0 * b8d, extra=4.0 * 2.3e6, share=0.4, per_step=4.0, dec_kernel="dec_layer_mf_kernel")
    e = events.kernel_entry(_args(), c)
    assert e["kernel"] == "dec_layer_mf_kernel" and e["bound"] == "hbm"
    assert e["algorithmic_bytes_per_launch"] == b8d and e["compulsory_bytes_per_launch"] == b8d + 2.3e6
    assert abs(e["achieved"] - b8d / 25e-6 / 1e9) < 1e-6 * e["achieved"] and abs(e["frac"] - e["achieved"] / 8000.0) < 1e-12
    assert abs(e["frac"] - 0.03995) < 1e-4  # 7.99 MB / 25 us = 320 GB/s
    # the bf16-arithmetic schedule's kernel is looked up under ITS name (round 4 read last round's dec_layer_mf_kernel pass for it)
    c1 = dict(c, dec_kernel="dec_layer_mf1_kernel")
    e1 = events.kernel_entry(_args(kv_bf16=True, attn_mfma=1), c1)
    assert e1["kernel"] == "dec_layer_mf1_kernel" and e1["bytes_per_pair"] == 529
    if e1.get("traffic_source"):
        assert e1["traffic_kernel"].startswith("dec_layer_mf1_kernel<")


def test_pmc_traffic_refuses_profiles_of_older_rounds(tmp_path, monkeypatch):
    from tools.benchlib import events

    prof = tmp_path / "profiles"
    prof.mkdir()
    wl = {"agents": 64, "polylines": 1024, "lights": 128, "scenes": 1, "rollouts": 1, "kv_bf16": False}
    k = {"dec_layer_mf_kernel<0,1>": {"launches": 10, "traffic_bytes_per_launch": 123}}
    (prof / "r04_c2_pmc.json").write_text(json.dumps({"workload": wl, "kernels": k}))
    monkeypatch.setattr(events, "ROOT", tmp_path)
    assert events.pmc_traffic(_args(), ["dec_layer_mf_kernel<"])[:2] == (123, "r04_c2_pmc.json")
    (prof / "r05_bench_line.json").write_text("{}")  # round 5 has evidence, but no PMC pass of this workload yet
    assert events.pmc_traffic(_args(), ["dec_layer_mf_kernel<"]) == (None, None, None, None)
    (prof / "r05_c2_pmc.json").write_text(json.dumps({"workload": wl, "kernels": k, "source_sha16": "abc"}))
    assert events.pmc_traffic(_args(), ["dec_layer_mf_kernel<"]) == (123, "r05_c2_pmc.json", "dec_layer_mf_kernel<0,1>", "abc")


def _rooflines(line, path=""):
    if isinstance(line, dict):
        for k, v in line.items():
            if k == "roofline" and isinstance(v, dict) and "frac" in v:
                yield path + "/roofline", v
            else:
                yield from _rooflines(v, path + "/" + k)


def test_committed_bench_line_rooflines_reproduce_from_their_own_fields():
    """Every `roofline` object of the newest committed judged line (profiles/rNN_bench_line.json, NN >= 5): frac = algorithmic bytes /
    average launch time / peak to 1e-3, and its PMC traffic comes from a profile of the SAME round."""
    import glob
    import re

    files = sorted(f for f in glob.glob(str(ROOT / "profiles" / "r??_bench_line.json")) if int(re.search(r"r(\d\d)_", f).group(1)) >= 5)
    if not files:
        import pytest

        pytest.skip("no judged line of round >= 5 committed yet")
    tag = re.search(r"(r\d\d)_", Path(files[-1]).name).group(1)
    line = json.load(open(files[-1]))
    n = 0
    for where, r in _rooflines(line):
        if r.get("bound") == "hbm" and r.get("algorithmic_bytes_per_launch") and r.get("avg_launch_us"):
            f = r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9 / r["peak"]
            assert abs(f - r["frac"]) <= 1e-3 * r["frac"], (where, f, r["frac"])
            n += 1
        if r.get("bound") == "mfma" and r.get("flops_per_launch") and r.get("avg_launch_us"):
            f = r["flops_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e12 / r["peak"]
            assert abs(f - r["frac"]) <= 1e-3 * r["frac"], (where, f, r["frac"])
            n += 1
        if r.get("traffic_source"):
            assert r["traffic_source"].startswith(tag + "_"), (where, r["traffic_source"])
    assert n >= 2
    head = line["roofline"]
    if head["kernel"].startswith("dec_layer_mf") and head["source_rows_per_launch"] == 64:
        assert head["algorithmic_bytes_per_launch"] == 7990400  # SURVEY 8d at configs[1]'s agents, nothing added
    if head["kernel"].startswith("dec_layer_mf") and "_pair_" in head["kernel"]:
        # the paired launch (Schedule.one_queue): the agents' 64 rows x 114 pairs + the lights' 128 rows x 48 pairs, two attention calls each
        assert head["source_rows_per_launch"] == 192 and abs(head["algorithmic_bytes_per_launch"] - (7990400 + 6922240)) <= 100


def _dry(argv, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv, "--dry-run"], env=env, cwd=str(ROOT), capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_eight_ranks_dry_run_shards_scenes_disjoint_and_covering():
    """`python bench.py --gpus 8 --dry-run`: the driver's SCALE invocation minus the GPU - bench.py starts 8 ranks itself, they
    rendezvous on 127.0.0.1 (gloo), every rank's scene ids are gathered: 8 contiguous disjoint blocks covering 0 .. 8 * S - 1
    (weak scaling: the data path needs no collective), the wall time is MAX-reduced over all 8."""
    line = _dry(["--gpus", "8", "--scenes", "3"])
    c = line["config"]
    assert line["n_gpus"] == 8 and c["max_rank_seen"] == 7
    assert c["scene_ids_per_rank"] == [[3 * r, 3 * r + 1, 3 * r + 2] for r in range(8)] and c["scene_ids_disjoint_and_covering"] is True


def test_eight_ranks_training_dry_run_broadcast_and_flat_allreduce():
    """`python bench.py --mode train --gpus 8 --dry-run` (configs[3] minus the GPU): 8 ranks with DIFFERENT initial weights end up with
    rank 0's after the flat broadcast, and the training step's one flat gradient all-reduce (FlatGrads over the default model's
    parameters, 42.6 MB here: without a backward pass nothing is excluded yet) returns the mean on every rank."""
    line = _dry(["--mode", "train", "--gpus", "8"])
    c = line["config"]
    assert line["n_gpus"] == 8 and line["unit"] == "scenes/s" and c["global_batch"] == 8 * 16 and c["parallelism"] == "dp8"
    assert c["scene_ids_disjoint_and_covering"] is True and len(c["scene_ids_per_rank"]) == 8
    assert c["checksum_equal_across_ranks"] is True and c["allreduce_is_the_mean_on_every_rank"] is True and c["rank_seeds_distinct"] is True
    assert 42_000_000 < c["allreduce_bytes"] <= 10657094 * 4 and c["broadcast_bytes"] >= c["allreduce_bytes"]  # (the trainable ones of 10,657,094)


# ---- round 6: the bound that binds the large-launch attention, the trace average beside the event average, size buckets
def test_attention_counters_give_the_valu_issue_bound():
    """on the big shapes `frac` stays SURVEY 8d's gathered-bytes model; `valu_issue_frac` = wave-wide VALU instructions per
    launch x 4 clocks / 1024 SIMDs / shader clock over the launch time, and `bound` says occupancy/latency once the counter-measured HBM
    traffic is under a quarter of the peak."""
    from tools.benchlib import events

    att = {"kernel": "knarpe_attn_kernel", "bound": "hbm", "frac": 0.74, "avg_launch_us": 41.7, "hbm_measured_frac": 0.13}
    cnt = {"valu_busy": 0.37, "l2_read_requests_per_launch": 710776.0, "valu_wave_insts_per_launch": 9323370.0, "counters_source": "r06_valu_attn_counters.json"}
    events.attach_attn_counters(att, cnt)
    assert abs(att["valu_issue_floor_us"] - 9323370.0 * 4 / 1024 / 2400.0) < 1e-9 and abs(att["valu_issue_floor_us"] - 15.17) < 0.01
    assert abs(att["valu_issue_frac"] - att["valu_issue_floor_us"] / 41.7) < 1e-12 and 0.36 < att["valu_issue_frac"] < 0.37
    assert att["bound"] == "occupancy/latency" and att["bound_8d"] == "hbm" and att["frac"] == 0.74 and "bound_note" in att
    assert abs(att["counters"]["l2_request_frac"] - 710776.0 * 128 / 41.7e-6 / 1e9 / events.L2_PEAK_GBS) < 1e-9
    # HBM-bound by the counters too (64 private scenes: 0.39 of the peak): the model's bound stays
    att2 = {"kernel": "knarpe_attn_kernel", "bound": "hbm", "frac": 0.5, "avg_launch_us": 42.0, "hbm_measured_frac": 0.39}
    events.attach_attn_counters(att2, dict(cnt))
    assert att2["bound"] == "hbm" and "bound_8d" not in att2 and "valu_issue_frac" in att2
    from tools.benchlib import report

    out = report.compact_roofline(att, sub=True)
    assert out["bound"] == "occupancy/latency" and out["bound_8d"] == "hbm" and "valu_issue_frac" in out and out["counters"]["valu_busy"] == 0.37


def test_trace_average_comes_from_this_rounds_kernel_stats(tmp_path, monkeypatch):
    """`avg_launch_us_trace` = the kernel's average launch in THIS round's committed `rocprofv3 --kernel-trace --stats` summary of the
    workload (the event pairs over-read by ~10 %: VERDICT r05 weak 2), the variant with the most launches; older rounds are not read."""
    from tools.benchlib import events

    prof = tmp_path / "profiles"
    prof.mkdir()
    row = "| `_ZN12_GLOBAL__N_119dec_layer_mf_kernelILb0ELb1EEEvNS_7MidArgsE.kd` | {} | 1566.7 | {} | 21.80 | 40.84 | 79.3 |\n"
    (prof / "r05_c2_kernel_stats.md").write_text("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n" + row.format(63760, 24.57))
    monkeypatch.setattr(events, "ROOT", tmp_path)
    assert events.trace_avg_us(_args(), "dec_layer_mf_kernel") == (24.57, "r05_c2_kernel_stats.md")
    (prof / "r06_sanitize_host.txt").write_text("x")  # round 6 has evidence but no kernel stats of this workload yet
    assert events.trace_avg_us(_args(), "dec_layer_mf_kernel") == (None, None)
    (prof / "r06_c2_kernel_stats.md").write_text(row.format(100, 99.0) + row.format(63000, 24.1) + "| `front_kernel` | 7924 | 108.2 | 13.66 | 12.84 | 29.84 | 5.5 |\n")
    assert events.trace_avg_us(_args(), "dec_layer_mf_kernel") == (24.1, "r06_c2_kernel_stats.md")
    assert events.trace_avg_us(_args(), "front_kernel") == (13.66, "r06_c2_kernel_stats.md")
    assert events.trace_avg_us(_args(agents=128, rollouts=32), "knarpe_attn_kernel") == (None, None)  # (another workload's file: r06_c5_...)
    c = dict(cls="dec_layer", key=64, t=4 * 27.5e-6, n=4, work=4.0 * 7990400, extra=0.0, share=0.4, per_step=4.0, dec_kernel="dec_layer_mf_kernel")
    e = events.kernel_entry(_args(), c)
    assert e["avg_launch_us_trace"] == 24.1 and e["trace_source"] == "r06_c2_kernel_stats.md"
    assert abs(e["frac_at_trace_avg"] - 7990400 / 24.1e-6 / 1e9 / 8000.0) < 1e-9 and e["frac_at_trace_avg"] > e["frac"]
