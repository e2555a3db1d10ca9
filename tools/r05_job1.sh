#!/bin/bash
# round-5 GPU job 1: re-run of the reduced closed-loop tests + the submission-shape test, a kernel trace of graph-replayed training steps
# (-> tools/train_replay_timeline.py), and the bench with the new shapes.
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out; cd $root
python -m pytest "tests/test_hip_rollout.py::test_teacher_forced_replay" "tests/test_hip_boundary.py::test_wosac_shape_joint_futures_vs_oracle" \
  "tests/test_hip_boundary.py::test_submission_shape_128_joint_futures_rule_checks_and_filter" -m gpu -q -s --no-header -p no:cacheprovider > $out/r05_tests_c.log 2>&1
tail -5 $out/r05_tests_c.log; grep -aE "vs oracle" $out/r05_tests_c.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp && cd $root
rocprofv3 --kernel-trace -d $out/kt_train -o kt -- python3 bench.py --mode train --no-cpu-baseline --steps 3 --warmup 2 --profile-steps 0 > $out/r05_train_replay.log 2>&1
db=$(find $out/kt_train -name '*.db' | head -1)
python3 tools/train_replay_timeline.py $db 30 > $out/r05a_train_replay_timeline.txt 2>&1
rm -rf $out/kt_train
tail -1 $out/r05_train_replay.log | cut -c1-400
head -60 $out/r05a_train_replay_timeline.txt
python bench.py --no-train-shape > $out/r05a_bench.log 2>&1
tail -1 $out/r05a_bench.log | cut -c1-6000
