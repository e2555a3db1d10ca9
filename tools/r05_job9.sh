#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out; cd $root
python -m pytest tests/test_hip_rules.py "tests/test_hip_boundary.py::test_wosac_shape_joint_futures_vs_oracle" "tests/test_hip_boundary.py::test_submission_shape_128_joint_futures_rule_checks_and_filter" -m gpu -q --no-header -p no:cacheprovider --durations=5 > $out/r05_tests_h.log 2>&1
tail -12 $out/r05_tests_h.log | cut -c1-200
( time python bench.py --no-train-shape ) > $out/r05g_bench.log 2>&1
grep -a '"metric"' $out/r05g_bench.log | tail -1 > $out/r05g_bench_line.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05g_bench_line.json"))
print(len(json.dumps(d)), d["value"], d["with_rule_checks"])
for k in ("wosac_shape","submission_shape","batched","bf16"):
    v=d[k]; print(k, v["value"], v.get("with_rule_checks"))
PY
tail -3 $out/r05g_bench.log | cut -c1-100
