#!/bin/bash
# steady-state step timelines (queue by queue) of the WOSAC shape and the submission shape, fp32 class
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out
for tag in c5:32 sub:128; do
  t=${tag%%:*}; r=${tag##*:}
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d /tmp/tl_$t -o tl -- python3 $root/bench.py --no-cpu-baseline --no-wosac-shape --agents 128 --rollouts $r --steps 40 --profile-steps 0 --new-scenes 0 > /dev/null 2>&1 )
  python3 $root/tools/step_timeline2.py $(ls /tmp/tl_$t/*.db | head -1) > $out/r05_${t}_two_stream_timeline.txt 2>&1
  rm -rf /tmp/tl_$t
done
cat $out/r05_c5_two_stream_timeline.txt | cut -c1-150
