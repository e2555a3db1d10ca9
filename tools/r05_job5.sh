#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out; cd $root
python -m pytest tests/test_hip_training.py -m gpu -q --no-header -p no:cacheprovider > $out/r05_tests_f.log 2>&1
tail -4 $out/r05_tests_f.log
python bench.py --mode train --no-cpu-baseline --train-precision bf16 --profile-steps 0 > $out/r05d_train_bf16.log 2>&1
tail -1 $out/r05d_train_bf16.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16', d['value'], d['ms_per_step'], d.get('loss'))"
python tools/train_op_table.py 45 > $out/r05d_train_op_table.txt 2>&1
grep -v "^$" $out/r05d_train_op_table.txt | cut -c1-230 | head -110
