#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out; cd $root
python -m pytest tests/test_hip_rules.py tests/test_hip_rollout.py "tests/test_hip_boundary.py::test_rollout_buffer_fields_vs_oracle_and_reference" "tests/test_hip_boundary.py::test_submission_shape_128_joint_futures_rule_checks_and_filter" -m gpu -q --no-header -p no:cacheprovider > $out/r05_tests_i.log 2>&1
tail -4 $out/r05_tests_i.log | cut -c1-200
for st in 20 80; do
python bench.py --steps $st --warmup 5 --no-train-shape --no-bf16-shape --no-submission-shape --no-batched-shape 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps $st', d['value'], d['with_rule_checks'], d['wosac_shape']['with_rule_checks'])"
done
