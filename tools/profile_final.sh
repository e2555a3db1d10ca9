# Final-build evidence for a round, on the GPU box (gpurun): GPU test suite, kernel-trace + PMC passes of the 64-agent scene and the
# 32 x 128 shape (tools/profile_round.sh), kernel stats of two eager training steps, and the default bench line. TAG=r01f by default;
# copy the summaries from gpurun_out/ into profiles/.
set -x
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/${TAG:-r01f}_gpu_tests.log
bash tools/profile_round.sh ${TAG:-r01f}_c2 64 1024 128 1 1
bash tools/profile_round.sh ${TAG:-r01f}_c5 128 1024 128 1 32 --steps 40
# the folded epilogue of the wave-per-row attention form (opt-in): bytes written per launch
TBX_ATTN_FOLD_BIG=1 bash tools/profile_round.sh ${TAG:-r01f}_c5_foldbig 128 1024 128 1 32 --steps 40
# one steady-state step of the default two-stream graph replay, kernel by kernel
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d /tmp/tl_final -o tl -- python3 $root/bench.py --no-cpu-baseline --no-wosac-shape --no-train-shape --no-bf16-shape --profile-steps 0 > /dev/null 2>&1 )
python3 tools/step_timeline2.py $(ls /tmp/tl_final/*.db | head -1) > gpurun_out/${TAG:-r01f}_c2_two_stream_timeline.txt 2>&1
rm -rf /tmp/tl_final
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $root
rocprofv3 --kernel-trace --stats -d $out/kt_train -o kt -- python3 bench.py --mode train --no-cpu-baseline --no-train-graph --steps 1 --warmup 1 > $out/${TAG:-r01f}_train_eager.log 2>&1
db=$(find $out/kt_train -name '*.db' | head -1)
{ echo "rocprofv3 --kernel-trace --stats -- python3 bench.py --mode train --no-cpu-baseline --no-train-graph --steps 1 --warmup 1   (2 eager training steps of 16 scenes; a graph replay takes the GPU-busy time of one)"; echo; tail -1 $out/${TAG:-r01f}_train_eager.log | cut -c1-300; echo; python3 tools/rocpd_stats.py $db 2>/dev/null | head -42; } > $out/${TAG:-r01f}_train_kernel_stats.md
rm -rf $out/kt_train
python bench.py > $out/${TAG:-r01f}_bench_default.log 2>&1
tail -1 $out/${TAG:-r01f}_bench_default.log | cut -c1-200
cat $out/${TAG:-r01f}_gpu_tests.log
