# usage: bash tools/scratch/ab_env.sh VAR [reps]   -> gpurun_out/ab_lines.txt: the C2 line with VAR=1 / VAR=0, alternating
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root; mkdir -p gpurun_out; rm -f gpurun_out/ab_lines.txt
for i in $(seq 1 ${2:-3}); do for v in 1 0; do
  echo -n "$1=$v " >> gpurun_out/ab_lines.txt
  env $1=$v python bench.py --no-cpu-baseline --no-wosac-shape --no-train-shape --no-bf16-shape --profile-steps 0 2>&1 | grep '"value"' | tail -1 | cut -c60-135 >> gpurun_out/ab_lines.txt
done; done
cat gpurun_out/ab_lines.txt
