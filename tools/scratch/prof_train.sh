set -x
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $root
rocprofv3 --kernel-trace --stats -d $out/kt_train -o kt -- python3 bench.py --mode train --no-cpu-baseline --no-train-graph --steps 1 --warmup 1 > $out/s6_train_eager.log 2>&1
db=$(find $out/kt_train -name '*.db' | head -1)
python3 tools/rocpd_stats.py $db | head -45 > $out/s6_train_eager_kernel_stats.md
rm -rf $out/kt_train
tail -1 $out/s6_train_eager.log | cut -c1-300
cat $out/s6_train_eager_kernel_stats.md | cut -c1-220
