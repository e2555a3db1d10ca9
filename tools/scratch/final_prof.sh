set -x
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/r01e_gpu_tests.log
bash tools/profile_round.sh r01e_c2 64 1024 128 1 1
bash tools/profile_round.sh r01e_c5 128 1024 128 1 32 --steps 40
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $root
rocprofv3 --kernel-trace --stats -d $out/kt_train -o kt -- python3 bench.py --mode train --no-cpu-baseline --no-train-graph --steps 1 --warmup 1 > $out/r01e_train_eager.log 2>&1
db=$(find $out/kt_train -name '*.db' | head -1)
{ echo "rocprofv3 --kernel-trace --stats -- python3 bench.py --mode train --no-cpu-baseline --no-train-graph --steps 1 --warmup 1   (2 eager training steps of 16 scenes; a graph replay takes the GPU-busy time of one)"; echo; tail -1 $out/r01e_train_eager.log | cut -c1-300; echo; python3 tools/rocpd_stats.py $db 2>/dev/null | head -42; } > $out/r01e_train_kernel_stats.md
rm -rf $out/kt_train
python bench.py > $out/r01e_bench_default.log 2>&1
tail -1 $out/r01e_bench_default.log | cut -c1-200
cat $out/r01e_gpu_tests.log
