#!/bin/bash
# MFMA-busy of the matrix-path kernels (tile_layer, tall_linear, dec_layer_mf, front, attn_mfma): ONE --pmc pass per workload
# (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE, kernel-trace only) at configs[1], the WOSAC
# shape (fp32 schedule and the reduced one) and two eager training steps -> gpurun_out/${TAG}_mfma_busy.json
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
tag=${TAG:-r04}
out=gpurun_out/pmc_busy; mkdir -p $out
common="--no-cpu-baseline --no-wosac-shape --no-lights-ahead --no-graph --profile-steps 0 --steps 6 --warmup 2 --new-scenes 0 --detail-file -"
declare -A W
W[c2]="bench.py $common"
W[c5]="bench.py $common --agents 128 --rollouts 32 --scenes 1"
W[c5_reduced]="bench.py $common --agents 128 --rollouts 32 --scenes 1 --kv-bf16 --attn-mfma 1"
W[train]="bench.py --mode train --no-cpu-baseline --no-train-graph --steps 1 --warmup 1 --profile-steps 0 --detail-file -"
echo "{" > gpurun_out/${tag}_mfma_busy.json
first=1
for w in c2 c5 c5_reduced train; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace -d $out/$w -o p -- python3 ${W[$w]} > $out/$w.log 2>&1
  db=$(find $out/$w -name '*.db' | head -1)
  [ $first = 1 ] || echo "," >> gpurun_out/${tag}_mfma_busy.json
  first=0
  echo "\"$w\": {\"cmd\": \"rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace -- python3 ${W[$w]}\", \"kernels\":" >> gpurun_out/${tag}_mfma_busy.json
  python3 tools/rocpd_counters.py "$db" >> gpurun_out/${tag}_mfma_busy.json
  echo "}" >> gpurun_out/${tag}_mfma_busy.json
  rm -rf $out/$w
done
echo "}" >> gpurun_out/${tag}_mfma_busy.json
python3 -c "
import json
d=json.load(open('gpurun_out/${tag}_mfma_busy.json'))
for w,v in d.items():
    print(w)
    for k,e in list(v['kernels'].items())[:10]:
        print('   %-44s x%-5d %8.1f us  mfma_busy %s' % (k, e['launches'], e['avg_launch_ns']/1e3, e.get('mfma_busy_frac_of_launch')))
"
