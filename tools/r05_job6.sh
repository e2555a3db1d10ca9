#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out; cd $root
python -m pytest tests/test_hip_training.py "tests/test_hip_parity.py::test_tl_tail_tile_equals_the_row_chain" tests/test_hip_rollout.py -m gpu -q --no-header -p no:cacheprovider > $out/r05_tests_g.log 2>&1
tail -6 $out/r05_tests_g.log
python bench.py --mode train --no-cpu-baseline --train-precision bf16 --profile-steps 0 > $out/r05e_train_bf16.log 2>&1
tail -1 $out/r05e_train_bf16.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16', d['value'], d['ms_per_step'], d.get('loss'))"
for sc in 16 64; do
python bench.py --scenes $sc --steps 40 --no-cpu-baseline --no-wosac-shape --profile-steps 0 --new-scenes 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('scenes $sc fp32', d['value'], d['ms_per_step'])"
python bench.py --scenes $sc --steps 40 --no-cpu-baseline --no-wosac-shape --profile-steps 0 --new-scenes 0 --kv-bf16 --attn-mfma 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('scenes $sc reduced', d['value'], d['ms_per_step'])"
done
