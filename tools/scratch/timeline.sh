# usage: TBX_...=.. bash tools/scratch/timeline.sh name  -> gpurun_out/timeline_<name>.txt (one steady-state step of the default C2 graph replay)
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out; rm -rf /tmp/tl_$1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d /tmp/tl_$1 -o tl -- python3 $root/bench.py --no-cpu-baseline --no-wosac-shape --no-train-shape --no-bf16-shape --profile-steps 0 --new-scenes 0 ${TL_ARGS} > /dev/null 2>&1 )
python3 $root/tools/step_timeline2.py $(ls /tmp/tl_$1/*.db | head -1) > $root/gpurun_out/timeline_$1.txt 2>&1
rm -rf /tmp/tl_$1
