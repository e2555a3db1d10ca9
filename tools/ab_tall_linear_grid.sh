#!/bin/bash
# tbx_tall_linear_bf16's persistent grid: one | two | three workgroups per CU (114 VGPRs, 40 KiB of LDS), training step bf16 class
out=gpurun_out; mkdir -p $out
for g in 256 512 768 256 512; do
  TBX_TALL_GRID=$g timeout 600 python bench.py --mode train --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tall_grid $g', d['value'], d['ms_per_step'])"
done > $out/ab_tall_linear_grid.txt 2>&1
cat $out/ab_tall_linear_grid.txt
