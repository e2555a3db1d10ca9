root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root; mkdir -p gpurun_out; out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $root
rm -rf $out/kt_train
rocprofv3 --kernel-trace -d $out/kt_train -o kt -- python3 bench.py --mode train --no-cpu-baseline --no-train-graph --steps 1 --warmup 1 --profile-steps 0 > $out/train_eager.log 2>&1
db=$(find $out/kt_train -name '*.db' | head -1)
python3 tools/scratch/train_seq.py $db knn_multi_kernel 130 > $out/train_seq.txt 2>&1
python3 tools/rocpd_stats.py $db 2>/dev/null | head -60 > $out/train_kernel_stats_now.md
rm -rf $out/kt_train
