#!/bin/bash
# round-5 GPU job 3: the bf16-class training kernels - their tests, the training step in both classes, a replay timeline of the bf16 class
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out; cd $root
python -m pytest tests/test_hip_training.py -m gpu -q -s --no-header -p no:cacheprovider -k "bf16 or mfma or training_step_vs or tall_linear_fn or wgrad" > $out/r05_tests_d.log 2>&1
tail -6 $out/r05_tests_d.log; grep -aE "^\.*\[(training|wgrad|tall)" $out/r05_tests_d.log | cut -c1-260
for prec in bf16 fp32; do
  python bench.py --mode train --no-cpu-baseline --train-precision $prec > $out/r05_train_$prec.log 2>&1
  tail -1 $out/r05_train_$prec.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$prec', d['value'], d['ms_per_step'], d.get('loss')); print(json.dumps(d.get('roofline'))[:600])"
done
cd /tmp && export TMPDIR=/tmp && cd $root
rocprofv3 --kernel-trace -d $out/kt_train -o kt -- python3 bench.py --mode train --no-cpu-baseline --steps 3 --warmup 2 --profile-steps 0 > $out/r05_train_replay_bf16.log 2>&1
db=$(find $out/kt_train -name '*.db' | head -1)
python3 tools/train_replay_timeline.py $db 30 > $out/r05b_train_replay_timeline_bf16.txt 2>&1
rm -rf $out/kt_train
head -40 $out/r05b_train_replay_timeline_bf16.txt
