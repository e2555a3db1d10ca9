#!/bin/bash
# round-5 GPU job 4: training after (a) VALU stepping attention, (c) windows from the chain kernel, (d) gathered gradients
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out; cd $root
python -m pytest tests/test_hip_training.py tests/test_hip_data_parallel.py -m gpu -q --no-header -p no:cacheprovider -x > $out/r05_tests_e.log 2>&1
tail -6 $out/r05_tests_e.log
for prec in bf16 fp32; do
  python bench.py --mode train --no-cpu-baseline --train-precision $prec --profile-steps 0 > $out/r05c_train_$prec.log 2>&1
  tail -1 $out/r05c_train_$prec.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$prec', d['value'], d['ms_per_step'], d.get('loss'))"
done
cd /tmp && export TMPDIR=/tmp && cd $root
rocprofv3 --kernel-trace -d $out/kt_train -o kt -- python3 bench.py --mode train --no-cpu-baseline --steps 3 --warmup 2 --profile-steps 0 > $out/r05c_train_replay_bf16.log 2>&1
db=$(find $out/kt_train -name '*.db' | head -1)
python3 tools/train_replay_timeline.py $db 40 > $out/r05c_train_replay_timeline_bf16.txt 2>&1
rm -rf $out/kt_train
head -48 $out/r05c_train_replay_timeline_bf16.txt
TBX_SPLIT_BF16=1 python bench.py --scenes 64 --steps 40 --no-cpu-baseline --no-wosac-shape --profile-steps 0 --new-scenes 0 2>/dev/null | tail -1 | cut -c1-200
python bench.py --scenes 64 --steps 40 --no-cpu-baseline --no-wosac-shape --profile-steps 0 --new-scenes 0 2>/dev/null | tail -1 | cut -c1-200
