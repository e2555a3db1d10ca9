root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root; mkdir -p gpurun_out
rm -f gpurun_out/ab_lines.txt
for v in 1 0; do
( cd /tmp && export TMPDIR=/tmp && export TBX_POOL_PROJ=$v && rocprofv3 --kernel-trace -d /tmp/tl_$v -o tl -- python3 $root/bench.py --no-cpu-baseline --no-wosac-shape --no-train-shape --no-bf16-shape --profile-steps 0 > /dev/null 2>&1 )
python3 tools/step_timeline2.py $(ls /tmp/tl_$v/*.db | head -1) > gpurun_out/tl_l16_$v.txt 2>&1
done
for v in 1 0 1 0; do TBX_POOL_PROJ=$v python bench.py --no-cpu-baseline --no-wosac-shape --no-train-shape --no-bf16-shape --profile-steps 0 2>&1 | tail -1 | cut -c60-130 >> gpurun_out/ab_lines.txt; done
