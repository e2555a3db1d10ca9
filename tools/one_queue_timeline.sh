# One steady-state step of the default (one-queue) graph replay at configs[1], kernel by kernel (rocprofv3 --kernel-trace + tools/step_timeline2.py)
F="--no-cpu-baseline --no-wosac-shape --no-train-shape --no-bf16-shape --no-rule-checks --no-submission-shape --no-batched-shape --profile-steps 0 --new-scenes 0"
root=$PWD
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d /tmp/tl_oq -o tl -- python3 $root/bench.py $F > /dev/null 2>&1 )
python3 tools/step_timeline2.py $(ls /tmp/tl_oq/*.db | head -1)
