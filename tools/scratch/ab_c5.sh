# usage: bash tools/scratch/ab_c5.sh reps "A=1" "A=0" ...  -> gpurun_out/ab_c5.txt: the WOSAC-shape (32 rollouts x 128 agents) line under each environment
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root; mkdir -p gpurun_out; rm -f gpurun_out/ab_c5.txt
reps=$1; shift
for i in $(seq 1 $reps); do for cfg in "$@"; do
  echo -n "[$cfg] " >> gpurun_out/ab_c5.txt
  env $cfg python bench.py --agents 128 --rollouts 32 --steps 40 --no-cpu-baseline --no-wosac-shape --no-train-shape --no-bf16-shape --profile-steps 0 --new-scenes 0 ${AB_ARGS} 2>&1 | grep '"value"' | tail -1 | cut -c60-135 >> gpurun_out/ab_c5.txt
done; done
cat gpurun_out/ab_c5.txt
