#!/bin/bash
# A/B of Schedule.attn_mfma at the WOSAC shape (32 rollouts x 128 agents): fp32 / bf16 tables x {VALU, bf16 MFMA}.
# Usage (GPU box): tools/ab_attn_mfma.sh <tag>   -> gpurun_out/<tag>_attn_ab.txt
tag=${1:-r04}
out=gpurun_out/${tag}_attn_ab.txt
: > $out
for kv in "" "--kv-bf16"; do
  for p in 0 1; do
    python3 bench.py --agents 128 --rollouts 32 --scenes 1 --steps 40 --no-cpu-baseline --new-scenes 0 $kv --attn-mfma $p --detail-file - 2>/dev/null | tail -1 |
      python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('kv=%-9s attn_mfma=%s  value %10.0f  ms/step %.4f  | %s %.1f us x%.0f share %.3f frac %.3f' % ('$kv' or 'fp32', $p, d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_us'], r['launches_per_step'], r['share_of_step_kernel_time'], r['frac']))" >> $out
  done
done
cat $out
