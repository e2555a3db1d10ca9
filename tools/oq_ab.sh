# A/B of the one-queue step (Schedule.one_queue) against the two-stream step at configs[1], and the one-queue step's kernel timeline
F="--no-cpu-baseline --no-wosac-shape --no-train-shape --no-bf16-shape --no-rule-checks --no-submission-shape --no-batched-shape --profile-steps 0 --new-scenes 0"
for v in ${AB:-1 0 1 0}; do echo "TBX_ONE_QUEUE=$v $(TBX_ONE_QUEUE=$v python bench.py $F $EXTRA 2>/dev/null | tail -1 | cut -c1-200)"; done
root=$PWD
( cd /tmp && export TMPDIR=/tmp && TBX_ONE_QUEUE=1 rocprofv3 --kernel-trace -d /tmp/tl_oq -o tl -- python3 $root/bench.py $F $EXTRA > /dev/null 2>&1 )
python3 tools/step_timeline2.py $(ls /tmp/tl_oq/*.db | head -1) > gpurun_out/${TAG:-r06}_c2_one_queue_timeline.txt 2>&1
cat gpurun_out/${TAG:-r06}_c2_one_queue_timeline.txt | head -20
