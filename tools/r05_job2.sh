#!/bin/bash
# round-5 GPU job 2: kernel-trace + PMC passes of the multi-scene serving shapes (16 and 64 scenes x 64 agents), fp32 and bf16 schedules,
# and a steady-state two-stream step timeline of the 64-scene shape.
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out; cd $root
bash tools/profile_round.sh r05_s16 64 1024 128 16 1 --steps 40 --new-scenes 0 --no-bf16-shape
bash tools/profile_round.sh r05_s64 64 1024 128 64 1 --steps 40 --new-scenes 0 --no-bf16-shape
bash tools/profile_round.sh r05_s64_bf16 64 1024 128 64 1 --steps 40 --new-scenes 0 --kv-bf16 --attn-mfma 1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d /tmp/tl_s64 -o tl -- python3 $root/bench.py --no-cpu-baseline --no-wosac-shape --scenes 64 --steps 40 --profile-steps 0 --new-scenes 0 > /dev/null 2>&1 )
python3 tools/step_timeline2.py $(ls /tmp/tl_s64/*.db | head -1) > $out/r05_s64_two_stream_timeline.txt 2>&1
rm -rf /tmp/tl_s64
head -40 $out/r05_s64_kernel_stats.md | cut -c1-200
tail -5 $out/r05_s64_two_stream_timeline.txt
