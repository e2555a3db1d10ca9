# usage: bash tools/scratch/ab_multi.sh reps "A=1 B=0" "A=0 B=0" ...   -> gpurun_out/ab_lines.txt: the C2 line under each environment, alternating
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root; mkdir -p gpurun_out; rm -f gpurun_out/ab_lines.txt
reps=$1; shift
for i in $(seq 1 $reps); do for cfg in "$@"; do
  echo -n "[$cfg] " >> gpurun_out/ab_lines.txt
  env $cfg python bench.py --no-cpu-baseline --no-wosac-shape --no-train-shape --no-bf16-shape --profile-steps 0 ${AB_ARGS} 2>&1 | grep '"value"' | tail -1 | cut -c60-135 >> gpurun_out/ab_lines.txt
done; done
cat gpurun_out/ab_lines.txt
