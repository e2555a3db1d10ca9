#!/bin/bash
# HBM traffic per launch of the kernels of an EAGER training step (configs[2]: 16 scenes): two rocprofv3 passes (--pmc FETCH_SIZE,
# --pmc WRITE_SIZE, kernel-trace only) of `bench.py --mode train --no-train-graph --steps 1 --warmup 1`, reduced by tools/rocpd_pmc.py
# to gpurun_out/${TAG}_train_pmc.json (copy it to profiles/: tools/benchlib/events.train_pmc_traffic reads it). TAG=r04 by default.
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
tag=${TAG:-r04}
out=$root/gpurun_out
cmd="bench.py --mode train --no-cpu-baseline --no-train-graph --steps 1 --warmup 1 --profile-steps 0"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace -d $out/pmc_train_$ctr -o p -- python3 $cmd > $out/${tag}_train_pmc_$ctr.log 2>&1
done
python3 tools/rocpd_pmc.py --agents 64 --polylines 1024 --lights 128 --scenes 16 --rollouts 1 --cmd "python3 $cmd" \
  $(find $out/pmc_train_FETCH_SIZE $out/pmc_train_WRITE_SIZE -name '*.db') > $out/${tag}_train_pmc.json
rm -rf $out/pmc_train_FETCH_SIZE $out/pmc_train_WRITE_SIZE
python3 - <<PY
import json
d = json.load(open("$out/${tag}_train_pmc.json"))
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"])[:12]:
    print(f"{k:44s} launches {v['launches']:5d}  {v['traffic_bytes_per_launch'] / 1e6:9.2f} MB per launch")
PY
